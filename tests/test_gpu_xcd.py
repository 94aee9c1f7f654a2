"""The small-N trial kernel (persistent=3; csrc/ldc_xcd_kernel.inc): a trial's ceil(M/16)^2 work-groups on ONE XCD, one
contraction family per wave over the full contraction index, resident operator fragments.

It is judged against the REFERENCE (golden fixtures: state <= 1e-12, scalars <= 1e-10, the reference's exact iteration
counts 59 649 / 273 012 / 41 261) and against the oracle -- not against the launch path: its contractions are single
accumulation chains instead of four K-quarters added in LDS, so it agrees with the launch path to rounding only.
"""
import json

import numpy as np
import pytest

from oracle import ldc_oracle as orc

pytestmark = pytest.mark.gpu


def make(N, Re, **kw):
    from solvers.spectral.sg import SGSolver
    args = dict(name="spectral", Re=float(Re), lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N,
                tolerance=1e-6, max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5,
                beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
                multigrid="none", check_every=512, graph_iters=16, persistent=3)
    args.update(kw)
    return SGSolver(**args)


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


def oracle_rows(o, K, diagnostics=True):
    rows = []
    for _ in range(K):
        up, vp = o.u.copy(), o.v.copy()
        dt = o.step()
        nrm = lambda a, b: np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12)      # noqa: E731
        rows.append([max(nrm(o.u, up), nrm(o.v, vp)), *o.residual_norms(), o.energy(),
                     o.enstrophy() if diagnostics else 0.0, o.palinstrophy() if diagnostics else 0.0, dt])
    return np.array(rows)


def uses_xcd(s):
    """The handle resolved to the small-N kernel (mode 3 is refused by the library where it does not apply)."""
    from solvers.spectral import ldc_lib as L
    return ((s.M + 15) // 16) ** 2 <= L.XCD_TILES


@pytest.mark.parametrize("N,Re,K", [(16, 100, 50), (32, 100, 500), (64, 400, 1000), (64, 1000, 3000)])
def test_xcd_trajectory_vs_reference(golden_dir, N, Re, K):
    """The reference's own K-step trajectories (g4), diagnostics on: one tile (N=16: no other work-group to wait for),
    9 tiles (N=32: T = 3 with index M-1 alone in the last tile row / column) and 25 tiles (N=64)."""
    g = np.load(golden_dir / f"g4_traj_N{N}_Re{Re}_K{K}.npz")
    s = make(N, Re)
    assert uses_xcd(s)
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
    # the post-processing reads what the kernel left behind (row-major + packed forms)
    assert rel(s._compute_vorticity(), g["omega"]) < 1e-10
    psi, _, _ = s._compute_streamfunction()
    assert np.max(np.abs(psi - g["psi"])) < 1e-10 * np.max(np.abs(g["psi"]))
    s.close()


@pytest.mark.parametrize("N,Re", [(8, 100), (12, 400), (15, 100), (20, 100), (24, 400), (30, 1000), (40, 100), (48, 400),
                                  (50, 1000), (72, 100), (79, 400)])
def test_xcd_records_vs_oracle_all_tilings(N, Re):
    """Every history column against the oracle for T = 1 ... 5 tiles per axis, sizes that are and are not multiples
    of 16 (the host's tail and non-tail layouts: the kernel re-tiles both as ceil(M/16)), the reference's Optuna
    sizes 30 / 40 / 50 and their FSG levels 15 / 20 / 24."""
    K = 40
    o = orc.OracleSG(N, Re)
    want = oracle_rows(o, K)
    s = make(N, Re)
    assert uses_xcd(s)
    rec = s.run_iterations(K)
    M = N + 1
    assert rec.shape == (K, 8)
    assert np.max(np.abs(s.arrays.u.reshape(M, M) - o.u)) < 1e-12
    assert np.max(np.abs(s.arrays.v.reshape(M, M) - o.v)) < 1e-12
    assert np.max(np.abs(s.arrays.p.reshape(M - 2, M - 2) - o.p)) < 1e-12
    assert rel(rec[:, 7], want[:, 7]) < 1e-12
    assert np.max(np.abs(rec[:, 0] - want[:, 0]) / (np.abs(want[:, 0]) + 1e-9)) < 1e-8
    for c in range(1, 7):
        assert rel(rec[:, c], want[:, c]) < (1e-10 if c < 5 else 1e-9), c
    s.close()


def test_xcd_larger_sizes_fall_back_to_the_launch_path():
    s = make(80, 100.0)                       # M = 81: 36 tiles do not fit the 32 CUs of an XCD
    assert not uses_xcd(s)
    rec = s.run_iterations(20)
    o = orc.OracleSG(80, 100.0)
    want = oracle_rows(o, 20)
    assert rel(rec[:, 7], want[:, 7]) < 1e-12 and np.max(np.abs(s.arrays.u.reshape(81, 81) - o.u)) < 1e-12
    s.close()


@pytest.mark.parametrize("N,K", [(16, 300), (32, 200), (64, 200)])
def test_xcd_smoother_mode_vs_oracle(N, K):
    """stage_pressure=1 (FSG levels): every stage differentiates its own stage pressure -- transforms, a second
    exchange and two more contractions per stage, all on waves 4-7."""
    from test_fsg import oracle_records
    Re = 1000.0
    s = make(N, Re, check_every=256)
    s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
    rec = s.run_iterations(K, diagnostics=False)
    o = orc.OracleSG(N, Re, stage_pressure=True)
    ref = oracle_records(o, K)
    assert rec.shape[0] == K and np.all(np.isfinite(rec[:, 0]))
    assert np.max(np.abs(s.arrays.u.reshape(N + 1, N + 1) - o.u)) < 1e-11
    assert np.max(np.abs(s.arrays.v.reshape(N + 1, N + 1) - o.v)) < 1e-11
    assert np.max(np.abs(s.arrays.p.reshape(N - 1, N - 1) - o.p)) < 1e-11
    assert rel(rec[:, 7], ref[:, 7]) < 1e-12
    for col in range(5):
        assert np.max(np.abs(rec[:, col] - ref[:, col]) / np.abs(ref[:, col])) < 1e-10, col
    s.close()


def test_xcd_solve_converges_like_reference(golden_dir):
    """Full solve() at N=32, Re=100, tol 1e-6 through the small-N kernel: the reference stops after 59 649 iterations."""
    meta = json.loads((golden_dir / "g7_converged_N32_Re100.json").read_text())["metrics"]
    g = np.load(golden_dir / "g7_converged_N32_Re100.npz")
    s = make(32, 100.0, check_every=2048)
    s.solve()
    m = s.metrics
    assert m.converged and m.iterations == meta["iterations"] == 59649
    assert np.max(np.abs(s.fields.u - g["u"])) < 1e-11
    assert np.max(np.abs(s.fields.v - g["v"])) < 1e-11
    assert np.max(np.abs(s.fields.p - g["p"])) < 1e-10
    for key in ("final_energy", "final_enstrophy", "final_palinstrophy", "psi_min", "psi_min_x", "psi_min_y"):
        assert getattr(m, key) == pytest.approx(meta[key], rel=1e-7, abs=1e-10), key
    for key in ("rel_iter_residual", "energy", "enstrophy", "palinstrophy"):
        assert rel(np.array(getattr(s.time_series, key)), g[f"ts_{key}"]) < 1e-8, key
    s.close()


def test_xcd_config2_converged_n64_re400_like_reference(golden_dir):
    """BASELINE config 2 (solver=spectral N=64 Re=400) to the reference's stopping rule through the small-N kernel:
    273 012 iterations exactly, fields, final metrics and all four histories of the reference's own run (g7b)."""
    meta = json.loads((golden_dir / "g7_converged_N64_Re400.json").read_text())
    ref, g = meta["metrics"], np.load(golden_dir / "g7_converged_N64_Re400.npz")
    s = make(64, 400.0, check_every=4096)
    s.solve()
    m = s.metrics
    assert m.converged and m.iterations == ref["iterations"] == 273012
    assert np.max(np.abs(s.fields.u - g["u"])) < 1e-10
    assert np.max(np.abs(s.fields.v - g["v"])) < 1e-10
    assert np.max(np.abs(s.fields.p - g["p"])) < 1e-9
    for key in ("final_energy", "final_enstrophy", "final_palinstrophy", "psi_min", "psi_min_x", "psi_min_y",
                "u_momentum_residual", "v_momentum_residual", "continuity_residual"):
        assert getattr(m, key) == pytest.approx(ref[key], rel=1e-7, abs=1e-10), key
    for key in ("rel_iter_residual", "energy", "enstrophy", "palinstrophy"):
        assert rel(np.array(getattr(s.time_series, key)), g[f"ts_{key}"]) < 1e-8, key
    s.close()


def test_xcd_fsg_runs_vs_reference(golden_dir):
    """FSG through the small-N kernel on every level that fits (capped runs of the reference incl. the new 32 -> 64 run
    and polynomial prolongation; the converged N=32 solve: 41 261 iterations exactly)."""
    from test_fsg import make_fsg
    g = np.load(golden_dir / "g8_fsg_runs.npz")
    meta = json.loads((golden_dir / "g8_fsg_runs.json").read_text())
    for name in ("cap300_N32_Re100", "cap200_N24_Re400", "lvl3_N48_Re100", "cap100_N64_Re1000", "poly_cap200_N32_Re400"):
        c = meta[name]
        s = make_fsg(c["N"], c["Re"], persistent=3, **c["kw"])
        s.solve()
        assert s.metrics.iterations == c["metrics"]["iterations"] and s.metrics.converged == c["metrics"]["converged"]
        for f in ("u", "v", "p"):
            assert np.max(np.abs(getattr(s.arrays, f) - g[f"{name}_{f}"])) < 1e-10, (name, f)
        s.close()
    ref = meta["full_N32_Re100"]["metrics"]
    s = make_fsg(32, 100.0, persistent=3)
    s.solve()
    assert s.metrics.converged and s.metrics.iterations == ref["iterations"] == 41261
    assert np.max(np.abs(s.fields.u - g["full_N32_Re100_u"])) < 1e-9
    assert s.metrics.psi_min == pytest.approx(ref["psi_min"], rel=1e-6)
    s.close()


def test_xcd_batch_runs_every_trial_on_an_xcd_of_its_own():
    """Eight N=64 trials (25 tiles each: one XCD per trial) and twenty N=24 trials (4 tiles: eight per XCD, so three
    launches' worth of slots are not needed -- one launch) advanced by ldc_batch_enqueue: each trial equals the same
    trial run alone through the same kernel bit for bit (the arithmetic does not depend on the placement), and the
    oracle to 1e-12."""
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    for N, B, K in ((64, 8, 300), (24, 20, 200), (40, 11, 150)):
        trials = [dict(name="spectral", Re=100.0 + 15.0 * q, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=0.0,
                       max_iterations=10**9, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                       corner_treatment="smoothing", corner_smoothing=0.05 + 0.005 * q, multigrid="none",
                       check_every=512, graph_iters=16, persistent=3) for q in range(B)]      # one chunk: same record boundaries
        b = BatchedSGSolver(trials)
        recs = b.run_iterations(K)
        for q in (0, B // 2, B - 1):
            one = SGSolver(**trials[q])
            r1 = one.run_iterations(K)
            assert np.array_equal(recs[q], r1), (N, q)
            assert np.array_equal(b.solvers[q].arrays.u, one.arrays.u) and np.array_equal(b.solvers[q].arrays.p, one.arrays.p)
            one.close()
        q = B - 1
        o = orc.OracleSG(N, trials[q]["Re"], corner_smoothing=trials[q]["corner_smoothing"])
        want = oracle_rows(o, K)
        assert np.max(np.abs(b.solvers[q].arrays.u.reshape(N + 1, N + 1) - o.u)) < 1e-12
        assert rel(recs[q][:, 7], want[:, 7]) < 1e-12 and rel(recs[q][:, 6], want[:, 6]) < 1e-9
        b.close()


def test_xcd_and_launch_path_hand_the_state_to_each_other():
    """The kernel reads phi^n from the row-major arrays and leaves row-major and packed forms (and, through the post
    launch that follows it, the pressure transforms) behind: chunks of the two paths can alternate on one state."""
    N, Re = 64, 400.0
    s = make(N, Re)
    rows = [s.run_iterations(100)]                     # 1 iteration launch path (edge fix) + 99 small-N kernel
    s.params.persistent = 0
    rows.append(s.run_iterations(37))                  # launch path (new handle) continues
    s.params.persistent = 3
    rows.append(s.run_iterations(64))
    s.params.persistent = 0
    rows.append(s.run_iterations(1))
    rec = np.concatenate(rows, axis=0)
    o = orc.OracleSG(N, Re)
    want = oracle_rows(o, 202)
    assert rec.shape == (202, 8)
    assert np.max(np.abs(s.arrays.u.reshape(N + 1, N + 1) - o.u)) < 1e-12
    assert np.max(np.abs(s.arrays.p.reshape(N - 1, N - 1) - o.p)) < 1e-12
    assert rel(rec[:, 7], want[:, 7]) < 1e-12
    for c in range(1, 7):
        assert rel(rec[:, c], want[:, c]) < (1e-10 if c < 5 else 1e-9), c
    s.close()
