"""GPU parity tests: the HIP path, called through the C ABI, against (a) golden vectors
from the reference and (b) the CPU oracle on identical inputs.

Tolerances (fp64): state after K steps <= 1e-12 abs (the measured noise floor
between two OpenBLAS kernels on the reference itself is 8e-16); scalars (dt, norms, E/Z/P)
<= 1e-10 rel; single residual evaluation <= 1e-11 rel to the field maximum.
"""
import json
import math

import numpy as np
import pytest

from oracle import ldc_oracle as orc

pytestmark = pytest.mark.gpu


def make(N, Re, **kw):
    from solvers.spectral.sg import SGSolver
    args = dict(name="spectral", Re=float(Re), lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N,
                tolerance=1e-6, max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5,
                beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
                multigrid="none", check_every=512, graph_iters=16)
    args.update(kw)
    return SGSolver(**args)


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def test_mfma_lane_maps():
    """Exact-integer A (16x4) and asymmetric B (4x16): D must equal A @ B element for element."""
    import torch
    from solvers.spectral import ldc_lib as L
    L.require_device()
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, size=(16, 4)).astype(float)
    B = (np.arange(64).reshape(4, 16) % 7 - 3.0) + 10.0 * np.arange(4)[:, None]
    dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    dD = torch.zeros((16, 16), dtype=torch.float64, device="cuda")
    L.check(L.lib().ldc_mfma_selftest(dA.data_ptr(), dB.data_ptr(), dD.data_ptr(), L.stream_ptr()))
    torch.cuda.synchronize()
    assert np.array_equal(dD.cpu().numpy(), A @ B)


def test_mfma_peak_benchmark_issues_at_the_pipe_rate():
    """`ldc_mfma_peak` (bench.py's `roofline.peak_measured`) must read the fp64 MFMA pipe, not its own code: built with launch
    bounds of 256 threads it kept its accumulators in AGPRs, copied them around every trip and reported 49 TFLOP/s for three
    rounds (one wave per SIMD: 140 cycles per MFMA instead of 64).  One wave per SIMD on every CU must come within 15 % of the
    datasheet's 78.6 TFLOP/s (measured: 76.6, tools/probes/mfma_rate_probe.hip)."""
    import torch
    from solvers.spectral import ldc_lib as L
    L.require_device()
    lib, st = L.lib(), L.stream_ptr()
    grid, iters = 256, 20000                       # 256 work-groups of four waves: one wave per SIMD
    sink = torch.zeros(grid * 256 + 2, dtype=torch.float64, device="cuda")
    L.check(lib.ldc_mfma_peak(sink.data_ptr(), 100, grid, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.check(lib.ldc_mfma_peak(sink.data_ptr(), iters, grid, st))
    e1.record()
    torch.cuda.synchronize()
    tflops = grid * 4 * iters * 8 * 2048.0 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    cycles = float(sink[0].item()) / (iters * 8)
    assert cycles < 70.0, f"{cycles:.1f} shader cycles per MFMA and wave (the pipe issues one per 64)"
    assert tflops > 0.85 * 78.6, f"{tflops:.1f} TFLOP/s"


@pytest.mark.parametrize("N", [16, 24])
def test_single_residual_vs_reference(golden_dir, N):
    g = np.load(golden_dir / "g3_single_stage.npz")
    s = make(N, 400.0)
    s.set_state(u=g[f"N{N}_u"], v=g[f"N{N}_v"], p=g[f"N{N}_p"])
    got = s.residual_fields()
    for key in ("du_dx", "du_dy", "dv_dx", "dv_dy", "lap_u", "lap_v", "dp_dx", "dp_dy", "R_u", "R_v", "R_p"):
        assert rel(got[key], g[f"N{N}_{key}"]) < 1e-11, key


@pytest.mark.parametrize("N", [20, 32, 47, 64, 80, 128])
def test_single_residual_vs_oracle(N):
    """Tail (N multiple of 16) and non-tail sizes, random smooth state, every intermediate.

    Two correct fp64 evaluations of sum_k a_k b_k differ by at most ~ n eps sum|a_k||b_k|;
    D2 has entries ~N^4/10 with heavy cancellation, so the bound is formed from |A| |B|
    (here with a constant of 4 eps, far below the worst case n eps)."""
    rng = np.random.default_rng(N)
    o = orc.OracleSG(N, 250.0)
    X, Y = np.meshgrid(o.ax.x, o.ay.x, indexing="ij")
    f = lambda: sum(rng.standard_normal() * np.sin((a + 1) * X + b * Y) for a in range(3) for b in range(3))
    o.u, o.v, o.p = f(), f(), f()[1:-1, 1:-1].copy()
    Ru, Rv, Rp, parts = o.residual(o.u, o.v, o.p, want_parts=True)
    s = make(N, 250.0)
    s.set_state(u=o.u, v=o.v, p=o.p)
    got = s.residual_fields()
    eps = np.finfo(float).eps
    A = np.abs
    Dx, Dy, D2x, D2y, Ix, Iy = o.ax.D, o.ay.D, o.ax.D2, o.ay.D2, o.ax.I, o.ay.I
    pf = A(Ix) @ A(o.p) @ A(Iy).T
    bound = dict(
        du_dx=A(Dx) @ A(o.u), du_dy=A(o.u) @ A(Dy).T, dv_dx=A(Dx) @ A(o.v), dv_dy=A(o.v) @ A(Dy).T,
        lap_u=A(D2x) @ A(o.u) + A(o.u) @ A(D2y).T, lap_v=A(D2x) @ A(o.v) + A(o.v) @ A(D2y).T,
        dp_dx=A(Dx) @ pf, dp_dy=pf @ A(Dy).T)
    nu = 1.0 / 250.0
    bound["R_u"] = A(o.u) * bound["du_dx"] + A(o.v) * bound["du_dy"] + bound["dp_dx"] + nu * bound["lap_u"]
    bound["R_v"] = A(o.u) * bound["dv_dx"] + A(o.v) * bound["dv_dy"] + bound["dp_dy"] + nu * bound["lap_v"]
    bound["R_p"] = (5.0 * (bound["du_dx"] + bound["dv_dy"]))[1:-1, 1:-1]
    want = dict(parts, R_u=Ru, R_v=Rv, R_p=Rp)
    # The pressure terms get a larger constant: the reference's inner-to-full interpolation matrices are unit rows on the
    # inner nodes only up to the rounding of V_full V_inner^-1 (|I - 1| <= 4e-15 = 16 eps at these sizes, measured), the
    # build sets those rows exactly (include/ldc_hip.h) -- a difference of a few eps |Dx| |p| in grad p.
    const = dict(dp_dx=24, dp_dy=24, R_u=24, R_v=24)
    for key, val in want.items():
        err = np.abs(got[key] - val.ravel())
        c = const.get(key, 4)
        assert np.all(err <= c * eps * bound[key].ravel() + 1e-300), (key, err.max(), (c * eps * bound[key]).max())


TRAJ = [(16, 100, 50), (32, 100, 500), (64, 400, 1000), (64, 1000, 3000)]


@pytest.mark.parametrize("N,Re,K", TRAJ)
def test_trajectory_vs_reference(golden_dir, N, Re, K):
    g = np.load(golden_dir / f"g4_traj_N{N}_Re{Re}_K{K}.npz")
    s = make(N, Re)
    rec = s.run_iterations(K)
    assert rec.shape == (K, 8)
    assert np.max(np.abs(s.arrays.u - g["u"])) < 1e-12
    assert np.max(np.abs(s.arrays.v - g["v"])) < 1e-12
    assert np.max(np.abs(s.arrays.p - g["p"])) < 1e-12
    assert rel(rec[:, 7], g["dt"]) < 1e-12
    assert np.max(np.abs(rec[:, 0] - g["rel"]) / (np.abs(g["rel"]) + 1e-9)) < 1e-8
    assert rel(rec[:, 1:4], g["res"]) < 1e-10
    assert rel(rec[:, 4], g["E"]) < 1e-10
    assert rel(rec[:, 5], g["Z"]) < 1e-10
    assert rel(rec[:, 6], g["P"]) < 1e-10
    # vorticity, stream function (reference: sparse LU) and the vortex table
    assert rel(s._compute_vorticity(), g["omega"]) < 1e-10
    psi, _, _ = s._compute_streamfunction()
    assert np.max(np.abs(psi - g["psi"])) < 1e-10 * np.max(np.abs(g["psi"]))
    vm = s.compute_vortex_metrics()
    for key, ref in zip(g["vortex_keys"], g["vortex_vals"]):
        assert abs(vm[str(key)] - ref) <= 1e-9 * max(abs(ref), 1.0), key


def compact_diffs(a, g, name):
    """Largest deviations of a 2-D array from the compact form a G4c fixture holds (make_golden._compact_state)."""
    rows, cols = g[f"{name}_rows_idx"], g[f"{name}_cols_idx"]
    d = max(np.max(np.abs(a[rows, :] - g[f"{name}_rows"])), np.max(np.abs(a[:, cols] - g[f"{name}_cols"])),
            np.max(np.abs(a[::8, ::8] - g[f"{name}_lattice"])))
    dn = abs(np.linalg.norm(a) - float(g[f"{name}_norm2"])) / max(float(g[f"{name}_norm2"]), 1e-300)
    dm = abs(np.max(np.abs(a)) - float(g[f"{name}_max"]))
    return d, dn, dm


def check_g4c(g, s, rec, N):
    """A solver's state and records against a G4c fixture (the reference's own run at N = 128 / 256)."""
    M = N + 1
    for name, a in (("u", s.arrays.u.reshape(M, M)), ("v", s.arrays.v.reshape(M, M)), ("p", s.arrays.p.reshape(M - 2, M - 2))):
        d, dn, dm = compact_diffs(a, g, name)
        assert d < 1e-12 and dn < 1e-12 and dm < 1e-12, (name, d, dn, dm)
    d, dn, _ = compact_diffs(np.asarray(s._compute_vorticity()).reshape(M, M), g, "omega")
    assert d < 1e-10 * float(g["omega_max"]) and dn < 1e-10, (d, dn)
    assert rel(rec[:, 7], g["dt"]) < 1e-12
    assert np.max(np.abs(rec[:, 0] - g["rel"]) / (np.abs(g["rel"]) + 1e-9)) < 1e-8
    assert rel(rec[:, 1:4], g["res"]) < 1e-10
    assert rel(rec[:, 4], g["E"]) < 1e-10
    assert rel(rec[:, 5], g["Z"]) < 1e-10
    assert rel(rec[:, 6], g["P"]) < 1e-10


G4C = [(128, 1000, 40), (256, 1000, 25), (128, 1000, 400), (256, 1000, 200)]


@pytest.mark.parametrize("N,Re,K", G4C)
def test_trajectory_at_headline_sizes_vs_reference(golden_dir, N, Re, K):
    """BASELINE configs 3-5 geometry against the REFERENCE's own runs (sg.py:410-449, base.py:250-276) on the launch
    path: every record, the end state on six rows / columns and the every-8th-node lattice, its norm and maximum."""
    g = np.load(golden_dir / f"g4c_traj_N{N}_Re{Re}_K{K}.npz")
    s = make(N, Re, persistent=0)
    rec = s.run_iterations(K)
    assert rec.shape == (K, 8)
    check_g4c(g, s, rec, N)


def test_variants_vs_reference(golden_dir):
    g = np.load(golden_dir / "g4b_variants.npz")
    meta = json.loads((golden_dir / "g4b_variants.json").read_text())
    for name, c in meta.items():
        s = make(c["N"], c["Re"], **c["kw"])
        rec = s.run_iterations(c["K"])
        assert np.max(np.abs(s.arrays.u - g[f"{name}_u"])) < 1e-12, name
        assert np.max(np.abs(s.arrays.v - g[f"{name}_v"])) < 1e-12, name
        assert np.max(np.abs(s.arrays.p - g[f"{name}_p"])) < 1e-12, name
        assert rel(rec[:, 7], g[f"{name}_dt"]) < 1e-12, name
        assert rel(rec[:, 6], g[f"{name}_P"]) < 1e-9, name


@pytest.mark.parametrize("tag", ["T16", "T32"])
def test_legendre_basis_vs_reference(golden_dir, tag):
    """basis_type='legendre': same device path, Legendre-Gauss-Lobatto operators from the host."""
    g = np.load(golden_dir / "g12_legendre.npz")
    c = json.loads((golden_dir / "g12_legendre.json").read_text())[tag]
    s = make(c["N"], c["Re"], basis_type="legendre")
    rec = s.run_iterations(c["K"])
    for k, a in (("u", s.arrays.u), ("v", s.arrays.v), ("p", s.arrays.p)):
        assert np.max(np.abs(a - g[f"{tag}_{k}"])) < 1e-11, k
    assert rel(rec[:, 7], g[f"{tag}_dt"]) < 1e-12
    assert rel(rec[:, 1:4], g[f"{tag}_res"]) < 1e-10
    for col, k in ((4, "E"), (5, "Z"), (6, "P")):
        assert rel(rec[:, col], g[f"{tag}_{k}"]) < 1e-9, k


def test_graph_replay_equals_eager_launches():
    """hipGraph replays and plain launches must give bit-identical trajectories."""
    a = make(32, 100.0, graph_iters=8, check_every=64)
    b = make(32, 100.0, graph_iters=4096, check_every=64)     # never reaches a full graph: eager
    ra, rb = a.run_iterations(64), b.run_iterations(64)
    assert np.array_equal(ra, rb)
    assert np.array_equal(a.arrays.u, b.arrays.u) and np.array_equal(a.arrays.p, b.arrays.p)


def test_runs_are_deterministic():
    a, b = make(64, 400.0), make(64, 400.0)
    ra, rb = a.run_iterations(200), b.run_iterations(200)
    assert np.array_equal(ra, rb) and np.array_equal(a.arrays.v, b.arrays.v)


def test_step_only_loop_matches_full_loop():
    """diagnostics=False (the step()-only rate) must not change the trajectory."""
    a, b = make(32, 100.0), make(32, 100.0)
    ra = a.run_iterations(100, diagnostics=True)
    rb = b.run_iterations(100, diagnostics=False)
    assert np.array_equal(a.arrays.u, b.arrays.u)
    assert np.array_equal(ra[:, :5], rb[:, :5]) and not rb[:, 5:7].any()


def test_n256_short_run_vs_oracle():
    """BASELINE config 3 geometry (N=256, Re=1000): 25 iterations against the oracle."""
    N, Re, K = 256, 1000.0, 25
    o = orc.OracleSG(N, Re)
    dts, Es, Ps = [], [], []
    for _ in range(K):
        dts.append(o.step()); Es.append(o.energy()); Ps.append(o.palinstrophy())
    s = make(N, Re)
    rec = s.run_iterations(K)
    M = N + 1
    assert np.max(np.abs(s.arrays.u.reshape(M, M) - o.u)) < 1e-12
    assert np.max(np.abs(s.arrays.v.reshape(M, M) - o.v)) < 1e-12
    assert np.max(np.abs(s.arrays.p.reshape(M - 2, M - 2) - o.p)) < 1e-12
    assert rel(rec[:, 7], np.array(dts)) < 1e-12
    assert rel(rec[:, 4], np.array(Es)) < 1e-10
    assert rel(rec[:, 6], np.array(Ps)) < 1e-9


@pytest.mark.parametrize("N,Re", [(8, 50.0), (12, 50.0), (48, 100.0), (96, 400.0), (100, 400.0), (272, 1000.0),
                                  (128, 1000.0), (160, 400.0), (200, 1000.0), (240, 100.0)])
def test_short_run_records_vs_oracle(N, Re):
    """Every history column against the oracle at sizes that place the index-(M-1) work differently:
    one tile holding all three jobs (N = 16 is in the fixtures), T = 3 (corner job on tile (0, 2)), T = 6,
    sizes that are no multiple of 16, and a multiple of 16 beyond 256 (index M-1 inside the tiles); T = 8 and 16
    take the XCD-patched tile order, T = 10, 13, 15 the plain one; 3 or 4 k-groups per wave."""
    K = 14 if N <= 100 else 12
    o = orc.OracleSG(N, Re)
    rows = []
    for _ in range(K):
        up, vp = o.u.copy(), o.v.copy()
        dt = o.step()
        nrm = lambda a, b: np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12)
        rows.append([max(nrm(o.u, up), nrm(o.v, vp)), *o.residual_norms(),
                     o.energy(), o.enstrophy(), o.palinstrophy(), dt])
    want = np.array(rows)
    s = make(N, Re)
    assert (s.tail == 1) == (N % 16 == 0 and N <= 256)
    rec = s.run_iterations(K)
    M = N + 1
    assert np.max(np.abs(s.arrays.u.reshape(M, M) - o.u)) < 1e-12
    assert np.max(np.abs(s.arrays.v.reshape(M, M) - o.v)) < 1e-12
    assert np.max(np.abs(s.arrays.p.reshape(M - 2, M - 2) - o.p)) < 1e-12
    assert rel(rec[:, 7], want[:, 7]) < 1e-12
    assert np.max(np.abs(rec[:, 0] - want[:, 0]) / (np.abs(want[:, 0]) + 1e-9)) < 1e-8
    for c in range(1, 7):
        assert rel(rec[:, c], want[:, c]) < (1e-10 if c < 5 else 1e-9), c


def test_solve_converges_like_reference(golden_dir):
    """Full solve() at N=32, Re=100, tol 1e-6: the reference stops after 59 649 iterations."""
    meta = json.loads((golden_dir / "g7_converged_N32_Re100.json").read_text())["metrics"]
    g = np.load(golden_dir / "g7_converged_N32_Re100.npz")
    s = make(32, 100.0, check_every=2048, graph_iters=32)
    s.solve()
    m = s.metrics
    assert m.converged and m.iterations == meta["iterations"] == 59649      # the reference's count, exactly
    assert np.max(np.abs(s.fields.u - g["u"])) < 1e-11
    assert np.max(np.abs(s.fields.v - g["v"])) < 1e-11
    assert np.max(np.abs(s.fields.p - g["p"])) < 1e-10
    assert abs(m.final_residual - meta["final_residual"]) < 1e-9 * meta["final_residual"] + 1e-15
    for key in ("final_energy", "final_enstrophy", "final_palinstrophy", "psi_min", "psi_min_x", "psi_min_y",
                "omega_center", "omega_max", "psi_BR", "psi_BL", "u_momentum_residual", "continuity_residual"):
        assert getattr(m, key) == pytest.approx(meta[key], rel=1e-7, abs=1e-10), key
    assert len(s.time_series.rel_iter_residual) == 1000
    # the FV comparison, through the reference's own compute_validation_errors in the fixture
    errs = s.compute_validation_errors()
    for key, ref in meta["validation_errors"].items():
        assert errs[key] == pytest.approx(ref, rel=1e-8), key
    g_err = s.ghia_error()
    assert 0 < g_err["u_rms"] < 0.02 and 0 < g_err["v_rms"] < 0.02
    # a second solve() on the converged state stops right after the warm-up
    s.solve(max_iter=50)
    assert s.metrics.iterations <= 12


def test_config2_converged_n64_re400_like_reference(golden_dir):
    """BASELINE config 2 -- solver=spectral N=64 Re=400 fp64 on one MI355X -- to the reference's stopping rule.
    The fixture is the reference's own run (tests/golden/make_golden.py --only G7b: 273 012 iterations, ten
    minutes of CPU there): same iteration count, fields, final metrics, histories and FV errors."""
    meta = json.loads((golden_dir / "g7_converged_N64_Re400.json").read_text())
    ref, g = meta["metrics"], np.load(golden_dir / "g7_converged_N64_Re400.npz")
    s = make(64, 400.0, check_every=4096, graph_iters=32)
    s.solve()
    m = s.metrics
    assert m.converged and m.iterations == ref["iterations"] == 273012
    assert np.max(np.abs(s.fields.u - g["u"])) < 1e-10
    assert np.max(np.abs(s.fields.v - g["v"])) < 1e-10
    assert np.max(np.abs(s.fields.p - g["p"])) < 1e-9
    assert abs(m.final_residual - ref["final_residual"]) < 1e-8 * ref["final_residual"]
    for key in ("final_energy", "final_enstrophy", "final_palinstrophy", "psi_min", "psi_min_x", "psi_min_y",
                "omega_center", "omega_max", "psi_BR", "psi_BL", "u_momentum_residual", "v_momentum_residual",
                "continuity_residual"):
        assert getattr(m, key) == pytest.approx(ref[key], rel=1e-7, abs=1e-10), key
    assert len(s.time_series.rel_iter_residual) == meta["time_series_len"]["rel_iter_residual"] == 1000
    for key in ("rel_iter_residual", "energy", "enstrophy", "palinstrophy"):
        got = np.array(getattr(s.time_series, key))
        assert rel(got, g[f"ts_{key}"]) < 1e-8, key
    errs = s.compute_validation_errors()
    for key, val in ref["validation_errors"].items():
        assert errs[key] == pytest.approx(val, rel=1e-8), key
    ghia = s.ghia_error()            # u_rms 0.0144, v_rms 0.0367 at this (N, Re)
    assert ghia["u_rms"] == pytest.approx(0.0144, abs=2e-4) and ghia["v_rms"] == pytest.approx(0.0367, abs=2e-4)


def test_max_iterations_and_nan_guard():
    s = make(32, 100.0)
    s.solve(max_iter=37)
    assert s.metrics.iterations == 37 and not s.metrics.converged
    assert len(s.time_series.rel_iter_residual) == 27
    # N=32, Re=1000 blows up with the default CFL; nan_guard exits early
    d = make(32, 1000.0, nan_guard=True, check_every=512)
    d.solve(max_iter=4000)
    assert not d.metrics.converged and d.metrics.iterations < 4000


def test_launcher_single_run_and_sweep(tmp_path, monkeypatch):
    """main.py: one run (the literal `solver=spectral` alias) and a 2x2 grid sweep, results gathered."""
    import importlib.util
    import json as _json
    from conftest import PKG
    spec = importlib.util.spec_from_file_location("ldc_main", PKG / "main.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.chdir(tmp_path)
    obj = mod.main(["solver=spectral", "N=16", "Re=100", "max_iterations=400"])
    runs = sorted(tmp_path.glob("hydra_outputs/multirun/*/*/0/results.json"))
    assert len(runs) == 1
    rec = _json.loads(runs[0].read_text())
    assert rec["run_name"] == "spectral_N17" and rec["metrics"]["iterations"] == 400
    assert rec["params"]["nx"] == 16 and rec["params"]["method"] == "Spectral-AC"
    assert set(rec["validation_errors"]) == {"u_L2_error", "v_L2_error"}
    assert obj == pytest.approx(math.hypot(rec["validation_errors"]["u_L2_error"], rec["validation_errors"]["v_L2_error"]))
    assert "ghia" in rec and (runs[0].parent / "solution.vts").exists()
    from solvers.vtkio import read_vts
    g = read_vts(runs[0].parent / "solution.vts")
    assert g["extent"] == (0, 16, 0, 16, 0, 0) and set(g["point_data"]) >= {"u", "v", "pressure", "vorticity", "velocity"}
    mod.main(["-m", "N=16,20", "Re=100,400", "max_iterations=60"])
    sweeps = sorted(tmp_path.glob("hydra_outputs/multirun/*/*/sweep_results.json"))
    recs = _json.loads(sweeps[-1].read_text())
    assert [(r["N"], r["Re"]) for r in recs] == [(16, 100), (16, 400), (20, 100), (20, 400)]
    assert all(r["metrics"]["iterations"] == 60 for r in recs)
