"""The persistent trial kernel (one launch per chunk of iterations, counter barrier per RK stage) against the
launch-per-stage path and against the reference's fixtures.

Both paths run the same arithmetic in the same order, so the bar between them is bit equality of every history
record and of the state (any stale read across a barrier would break it); against the reference the tolerances
are those of tests/test_gpu_parity.py.  Sizes cover one tile (T = 1), index M-1 outside the tiles (N = 32, 48,
64, 96, 128) and inside them (N = 24, 40, 100), with and without the fused diagnostics, and the smoother mode of
the FSG levels."""
import json

import numpy as np
import pytest

from test_gpu_parity import make, rel

pytestmark = pytest.mark.gpu


def _run(N, Re, K, persistent, diagnostics=True, **kw):
    s = make(N, Re, persistent=persistent, check_every=256, **kw)
    rec = s.run_iterations(K, diagnostics=diagnostics)
    out = (rec, s.arrays.u.copy(), s.arrays.v.copy(), s.arrays.p.copy())
    s.close()
    return out


@pytest.mark.parametrize("N,K", [(16, 300), (24, 300), (32, 700), (40, 300), (48, 300), (64, 700), (96, 300),
                                 (100, 300), (128, 300)])
@pytest.mark.parametrize("diagnostics", [True, False])
def test_persistent_is_bit_identical_to_launch_per_stage(N, K, diagnostics):
    a = _run(N, 400.0, K, 1, diagnostics)
    b = _run(N, 400.0, K, 0, diagnostics)
    assert a[0].shape == (K, 8) and np.all(np.isfinite(a[0]))
    for x, y, name in zip(a, b, ("records", "u", "v", "p")):
        assert np.array_equal(x, y), (name, float(np.max(np.abs(x - y))))


@pytest.mark.parametrize("N,K", [(16, 300), (24, 300), (32, 700), (40, 300), (48, 300), (64, 700), (72, 300), (80, 300)])
@pytest.mark.parametrize("diagnostics", [True, False])
def test_one_xcd_placement_is_bit_identical_to_launch_per_stage(N, K, diagnostics):
    """persistent=2: the trial's work-groups claim their tiles on ONE XCD and exchange state through its L2 (plain
    stores, L1-bypassing loads).  Same arithmetic, so again bit equality -- a stale line would break it."""
    a = _run(N, 400.0, K, 2, diagnostics)
    b = _run(N, 400.0, K, 0, diagnostics)
    assert a[0].shape == (K, 8) and np.all(np.isfinite(a[0]))
    for x, y, name in zip(a, b, ("records", "u", "v", "p")):
        assert np.array_equal(x, y), (name, float(np.max(np.abs(x - y))))


def test_one_xcd_placement_trajectory_vs_reference(golden_dir):
    g = np.load(golden_dir / "g4_traj_N64_Re400_K1000.npz")
    s = make(64, 400, persistent=2)
    rec = s.run_iterations(1000)
    for name in ("u", "v", "p"):
        assert np.max(np.abs(getattr(s.arrays, name) - g[name])) < 1e-12, name
    assert rel(rec[:, 7], g["dt"]) < 1e-12 and rel(rec[:, 1:4], g["res"]) < 1e-10
    s.close()


def test_persistent_chunking_does_not_matter():
    """700 iterations in chunks of 256 and in chunks of 37: same state and same records, bit for bit -- except
    Z and P of the LAST record of a chunk, which the closing stand-alone omega / palinstrophy kernels compute
    (nothing follows that would carry them); those agree to rounding."""
    ref = _run(64, 1000.0, 700, 1)
    s = make(64, 1000.0, persistent=1, check_every=37)
    rec = s.run_iterations(700)
    assert np.array_equal(s.arrays.u, ref[1]) and np.array_equal(s.arrays.v, ref[2]) and np.array_equal(s.arrays.p, ref[3])
    keep = [0, 1, 2, 3, 4, 7]
    assert np.array_equal(rec[:, keep], ref[0][:, keep])
    assert rel(rec[:, 5], ref[0][:, 5]) < 1e-12 and rel(rec[:, 6], ref[0][:, 6]) < 1e-12
    inner = np.ones(700, bool)
    inner[np.arange(36, 700, 37)] = False
    inner[[255, 511, 699]] = False
    assert np.array_equal(rec[inner][:, 5:7], ref[0][inner][:, 5:7])


@pytest.mark.parametrize("N,Re,K", [(32, 100, 500), (64, 400, 1000)])
def test_persistent_trajectory_vs_reference(golden_dir, N, Re, K):
    g = np.load(golden_dir / f"g4_traj_N{N}_Re{Re}_K{K}.npz")
    s = make(N, Re, persistent=1)
    rec = s.run_iterations(K)
    assert np.max(np.abs(s.arrays.u - g["u"])) < 1e-12
    assert np.max(np.abs(s.arrays.v - g["v"])) < 1e-12
    assert np.max(np.abs(s.arrays.p - g["p"])) < 1e-12
    assert rel(rec[:, 7], g["dt"]) < 1e-12
    assert rel(rec[:, 1:4], g["res"]) < 1e-10
    for col, key in ((4, "E"), (5, "Z"), (6, "P")):
        assert rel(rec[:, col], g[key]) < 1e-10, key


def test_persistent_solve_stops_where_the_reference_does(golden_dir):
    meta = json.loads((golden_dir / "g7_converged_N32_Re100.json").read_text())["metrics"]
    g = np.load(golden_dir / "g7_converged_N32_Re100.npz")
    s = make(32, 100.0, persistent=1, check_every=4096)
    s.solve()
    assert s.metrics.converged and s.metrics.iterations == meta["iterations"]
    assert np.max(np.abs(s.fields.u - g["u"])) < 1e-10 and np.max(np.abs(s.fields.p - g["p"])) < 1e-10


@pytest.mark.parametrize("N,levels", [(32, 2), (48, 3), (64, 2)])
def test_persistent_smoother_mode_matches_launch_path(N, levels):
    """FSG: every level in smoother mode (a transform phase after each stage) -- persistent == launches, bit for bit."""
    from solvers.spectral.fsg import FSGSolver
    out = []
    for persistent in (1, 0, 2):        # 2: one-XCD placement on the levels that fit an XCD, launches on the others
        s = FSGSolver(name="spectral_fsg", Re=400.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                      corner_treatment="smoothing", corner_smoothing=0.15, multigrid="fsg", n_levels=levels,
                      coarse_tolerance_factor=10.0, tolerance=1e-6, max_iterations=400, check_every=128,
                      graph_iters=16, persistent=persistent)
        s.solve()
        out.append((s.metrics.iterations, s.fields.u.copy(), s.fields.v.copy(), s.fields.p.copy(),
                    s.metrics.final_energy, s.metrics.final_enstrophy))
        s.close()
    assert out[0][0] == out[1][0] == out[2][0]
    for x, y, z in zip(out[0][1:], out[1][1:], out[2][1:]):
        assert np.array_equal(x, y) and np.array_equal(z, y)


def test_persistent_mode_switch_and_limits():
    from solvers.spectral import ldc_lib as L
    s = make(32, 100.0, persistent=1)
    s._begin(0.0)
    assert L.lib().ldc_solver_status(s._handle) == 0
    assert L.lib().ldc_solver_set_persistent(s._handle, 7) == -1          # LDC_E_ARG
    assert L.lib().ldc_solver_set_persistent(s._handle, 2) == 0           # 4 tiles: fits one XCD
    assert L.lib().ldc_solver_set_persistent(s._handle, 0) == 0
    s.close()
    mid = make(96, 100.0)                       # 36 tiles: persistent yes, on one XCD (32 CUs) no
    mid._begin(0.0)
    assert L.lib().ldc_solver_set_persistent(mid._handle, 1) == 0
    assert L.lib().ldc_solver_set_persistent(mid._handle, 2) == -1
    mid.close()
    big = make(272, 100.0)                      # 17 x 17 = 289 work-groups: more than the chip has CUs
    big._begin(0.0)
    assert L.lib().ldc_solver_set_persistent(big._handle, 1) == -1
    big.close()


@pytest.mark.parametrize("kw", [dict(Lx=2.0, Ly=1.0, lid_velocity=1.5), dict(corner_treatment="saad"),
                                dict(basis_type="legendre"), dict(CFL=0.8, beta_squared=2.0, corner_smoothing=0.05)],
                         ids=["rect", "saad", "legendre", "cfl"])
def test_persistent_variants_bit_identical(kw):
    """Non-default parameters of the reference's constructor (rectangular cavity, Saad lid, Legendre basis, other
    CFL / beta^2): the persistent kernel and the launch path agree bit for bit there too."""
    a = _run(48, 100.0, 200, 1, True, **kw)
    b = _run(48, 100.0, 200, 0, True, **kw)
    for x, y, name in zip(a, b, ("records", "u", "v", "p")):
        assert np.array_equal(x, y), name
    assert np.all(np.isfinite(a[0]))
