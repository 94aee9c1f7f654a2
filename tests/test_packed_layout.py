"""The packed twins (include/ldc_hip.h, Conventions): layout pinned on the host, ldc_pack and the twins the
kernels keep in step pinned on the GPU.  A twin that drifted from its row-major array would still give a
self-consistent but WRONG trajectory, so this is checked directly and not only through the trajectories."""
import numpy as np
import pytest


def pack_numpy(a: np.ndarray) -> np.ndarray:
    """Packed twin of a row-major LD x LD array, written from the header's description: block (R, G) of 16 x 16 at
    (R * NB + G) * 256, element (r, 4c + s) of the block at (16c + r) * 4 + s."""
    LD = a.shape[0]
    assert a.shape == (LD, LD) and LD % 16 == 0
    NB = LD // 16
    out = np.empty(LD * LD)
    for R in range(NB):
        for G in range(NB):
            blk = a[16 * R: 16 * R + 16, 16 * G: 16 * G + 16]          # [r][k]
            lanes = blk.reshape(16, 4, 4).transpose(1, 0, 2)            # [c][r][s]
            out[(R * NB + G) * 256: (R * NB + G + 1) * 256] = lanes.ravel()
    return out


def test_pack_numpy_matches_the_formula():
    LD = 48
    a = np.arange(LD * LD, dtype=float).reshape(LD, LD)
    p = pack_numpy(a)
    NB = LD // 16
    rng = np.random.default_rng(0)
    for _ in range(500):
        i, k = rng.integers(0, LD, size=2)
        R, r, G, kk = i // 16, i % 16, k // 16, k % 16
        c, s = kk // 4, kk % 4
        assert p[(R * NB + G) * 256 + (16 * c + r) * 4 + s] == a[i, k]
    assert sorted(p) == sorted(a.ravel())            # a permutation
    # lane l of a wave (row l & 15, k-chunk l >> 4) finds its four k-steps contiguously at 4 l
    blk = p[:256].reshape(64, 4)
    for l in (0, 5, 17, 63):
        assert np.array_equal(blk[l], a[l & 15, 4 * (l >> 4): 4 * (l >> 4) + 4])


@pytest.mark.gpu
@pytest.mark.parametrize("LD", [16, 48, 272])
def test_ldc_pack_matches_numpy(LD):
    import torch
    from solvers.spectral import ldc_lib as L
    L.require_device()
    rng = np.random.default_rng(LD)
    a = rng.standard_normal((LD, LD))
    src = torch.from_numpy(a).cuda()
    dst = torch.zeros(LD * LD, dtype=torch.float64, device="cuda")
    L.check(L.lib().ldc_pack(src.data_ptr(), dst.data_ptr(), LD, L.stream_ptr()), "ldc_pack")
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), pack_numpy(a))
    assert L.lib().ldc_pack(src.data_ptr(), src.data_ptr(), LD, L.stream_ptr()) == -1     # LDC_E_ARG: in place
    assert L.lib().ldc_pack(src.data_ptr(), dst.data_ptr(), LD + 1, L.stream_ptr()) != 0


def _twins_in_step(s, names):
    """Every block an MFMA operand load can touch (R, G < T) of each twin equals the packed row-major array."""
    T, NB, LD = s.T, s.LD // 16, s.LD
    bad = []
    for n in names:
        want = pack_numpy(s.d[n].cpu().numpy()).reshape(NB, NB, 256)[:T, :T]
        got = s.d[n + "K"].cpu().numpy().reshape(NB, NB, 256)[:T, :T]
        if not np.array_equal(want, got):
            bad.append((n, float(np.max(np.abs(want - got)))))
    return bad


@pytest.mark.gpu
@pytest.mark.parametrize("N,diag", [(32, True), (40, True), (64, False), (100, True)])
def test_kernels_keep_twins_in_step(N, diag):
    """After real iterations (graph replays and eager launches) the state, stage-buffer and pressure-transform
    twins still mirror their row-major arrays bit for bit; operators were packed by the host."""
    import torch
    from solvers.spectral.sg import SGSolver
    s = SGSolver(name="spectral", Re=400.0, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=1e-6,
                 max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                 corner_treatment="smoothing", corner_smoothing=0.15, multigrid="none", check_every=64, graph_iters=8)
    assert not _twins_in_step(s, ("Dx", "D2x", "Dy", "D2y", "IxF", "GxF", "IyF", "GyF", "U", "UT", "V", "VT", "P"))
    s.run_iterations(21, diagnostics=diag)       # 2 graph replays + 5 eager iterations
    torch.cuda.synchronize()
    # (the velocity stage buffers UA.. / UB.. exist in packed form only: include/ldc_hip.h)
    state = ("U", "UT", "V", "VT", "P", "T1T", "T2T")
    assert not _twins_in_step(s, state)
    # and an upload re-packs
    u = np.random.default_rng(1).standard_normal((N + 1, N + 1))
    s.set_state(u=u)
    assert not _twins_in_step(s, ("U", "UT", "UA", "UAT", "UB", "UBT"))
    s.close()
