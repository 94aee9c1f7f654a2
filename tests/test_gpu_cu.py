"""The trial-per-CU kernel (persistent=4; csrc/ldc_cu_kernel.inc): ONE work-group advances a trial -- stage state and
operators in the CU's LDS, one wave per 16 x 16 tile, no hand-over between work-groups; a batch is one launch with one
work-group per trial.

Judged like the small-N kernel (tests/test_gpu_xcd.py): against the REFERENCE (golden fixtures: state <= 1e-12, the exact
iteration counts 59 649 / 41 261) and against the oracle; it agrees with the other GPU paths to rounding only.
"""
import json

import numpy as np
import pytest

from oracle import ldc_oracle as orc
from test_gpu_xcd import oracle_rows, rel

pytestmark = pytest.mark.gpu


def make(N, Re, **kw):
    from solvers.spectral.sg import SGSolver
    args = dict(name="spectral", Re=float(Re), lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N,
                tolerance=1e-6, max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5,
                beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
                multigrid="none", check_every=512, graph_iters=16, persistent=4)
    args.update(kw)
    return SGSolver(**args)


def mode_of(s):
    from solvers.spectral import ldc_lib as L
    if s._handle is None:
        s._begin(s.params.tolerance)
    return int(L.lib().ldc_solver_mode(s._handle))


@pytest.mark.parametrize("N,Re,K", [(16, 100, 50), (32, 100, 500)])
def test_cu_trajectory_vs_reference(golden_dir, N, Re, K):
    """The reference's own K-step trajectories (g4), diagnostics on: 4 waves (N=16) and 9 waves (N=32: index M-1 alone in
    the last tile row / column)."""
    g = np.load(golden_dir / f"g4_traj_N{N}_Re{Re}_K{K}.npz")
    s = make(N, Re)
    rec = s.run_iterations(K)
    assert mode_of(s) == 4
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


@pytest.mark.parametrize("N,Re", [(8, 100), (12, 400), (15, 100), (16, 400), (20, 100), (24, 400), (30, 1000), (31, 100),
                                  (32, 400), (33, 100), (36, 100), (39, 400), (40, 100), (41, 100), (43, 400)])
@pytest.mark.parametrize("diagnostics", [True, False])
def test_cu_records_vs_oracle_all_sizes(N, Re, diagnostics):
    """Every history column against the oracle for 1, 4 and 9 waves, every residue of M mod 4 (the contraction range is
    ceil(M / 4) k-steps, zero padded) and both row strides of the LDS arrays, the reference's Optuna sizes 30 / 40 and their
    FSG levels 15 / 20, up to the largest size the LDS holds (M = 44)."""
    K = 40
    o = orc.OracleSG(N, Re)
    want = oracle_rows(o, K, diagnostics)
    s = make(N, Re)
    rec = s.run_iterations(K, diagnostics=diagnostics)
    assert mode_of(s) == 4
    M = N + 1
    assert rec.shape == (K, 8)
    assert np.max(np.abs(s.arrays.u.reshape(M, M) - o.u)) < 1e-12
    assert np.max(np.abs(s.arrays.v.reshape(M, M) - o.v)) < 1e-12
    assert np.max(np.abs(s.arrays.p.reshape(M - 2, M - 2) - o.p)) < 1e-12
    assert rel(rec[:, 7], want[:, 7]) < 1e-12
    assert np.max(np.abs(rec[:, 0] - want[:, 0]) / (np.abs(want[:, 0]) + 1e-9)) < 1e-8
    for c in range(1, 7 if diagnostics else 5):
        assert rel(rec[:, c], want[:, c]) < (1e-10 if c < 5 else 1e-9), c
    s.close()


def test_cu_larger_sizes_fall_back_to_the_launch_path():
    s = make(44, 100.0)                       # M = 45: ten arrays of 48 x 50 doubles do not fit 160 KB
    rec = s.run_iterations(20)
    assert mode_of(s) == 0
    o = orc.OracleSG(44, 100.0)
    want = oracle_rows(o, 20)
    assert rel(rec[:, 7], want[:, 7]) < 1e-12 and np.max(np.abs(s.arrays.u.reshape(45, 45) - o.u)) < 1e-12
    s.close()


@pytest.mark.parametrize("N,K", [(15, 300), (16, 300), (32, 200), (40, 100)])
def test_cu_smoother_mode_vs_oracle(N, K):
    """stage_pressure=1 (FSG levels): every stage differentiates its own stage pressure; the ring vectors are formed
    behind stage 3."""
    from test_fsg import oracle_records
    Re = 1000.0
    s = make(N, Re, check_every=256)
    s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
    rec = s.run_iterations(K, diagnostics=False)
    assert mode_of(s) == 4
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


def test_cu_solve_converges_like_reference(golden_dir):
    """Full solve() at N=32, Re=100, tol 1e-6 on one CU: the reference stops after 59 649 iterations."""
    meta = json.loads((golden_dir / "g7_converged_N32_Re100.json").read_text())["metrics"]
    g = np.load(golden_dir / "g7_converged_N32_Re100.npz")
    s = make(32, 100.0, check_every=4096)
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


def test_cu_fsg_runs_vs_reference(golden_dir):
    """FSG with every level on one CU (capped runs of the reference incl. polynomial prolongation; the converged N=32
    solve: 41 261 iterations exactly)."""
    from test_fsg import make_fsg
    g = np.load(golden_dir / "g8_fsg_runs.npz")
    meta = json.loads((golden_dir / "g8_fsg_runs.json").read_text())
    for name in ("cap300_N32_Re100", "cap200_N24_Re400", "poly_cap200_N32_Re400"):
        c = meta[name]
        s = make_fsg(c["N"], c["Re"], persistent=4, **c["kw"])
        s.solve()
        assert s.metrics.iterations == c["metrics"]["iterations"] and s.metrics.converged == c["metrics"]["converged"]
        for f in ("u", "v", "p"):
            assert np.max(np.abs(getattr(s.arrays, f) - g[f"{name}_{f}"])) < 1e-10, (name, f)
        s.close()
    ref = meta["full_N32_Re100"]["metrics"]
    s = make_fsg(32, 100.0, persistent=4)
    s.solve()
    assert s.metrics.converged and s.metrics.iterations == ref["iterations"] == 41261
    assert np.max(np.abs(s.fields.u - g["full_N32_Re100_u"])) < 1e-9
    assert s.metrics.psi_min == pytest.approx(ref["psi_min"], rel=1e-6)
    s.close()


@pytest.mark.parametrize("N,B,K", [(32, 40, 200), (16, 300, 100), (40, 9, 100)])
def test_cu_batch_is_one_launch_with_a_work_group_per_trial(N, B, K):
    """B trials advanced by ldc_batch_enqueue in one launch (300 trials: more work-groups than the chip has CUs -- the
    later ones start when earlier ones end, nobody waits for anybody): each trial equals the same trial run alone through
    the same kernel bit for bit, and the oracle to 1e-12."""
    from solvers.spectral import ldc_lib as L
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    trials = [dict(name="spectral", Re=100.0 + 3.0 * q, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=0.0,
                   max_iterations=10**9, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                   corner_treatment="smoothing", corner_smoothing=0.05 + 0.0005 * q, multigrid="none",
                   check_every=512, graph_iters=16, persistent=4) for q in range(B)]
    b = BatchedSGSolver(trials)
    recs = b.run_iterations(K)
    assert all(L.lib().ldc_solver_mode(s._handle) == 4 for s in b.solvers)
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


def test_cu_and_the_other_paths_hand_the_state_to_each_other():
    """The kernel reads phi^n from the row-major arrays and leaves row-major and packed forms (and, through the post
    launch that follows it, the pressure transforms) behind: chunks of the three paths can alternate on one state."""
    N, Re = 32, 400.0
    s = make(N, Re)
    rows = [s.run_iterations(100)]                     # 1 iteration launch path (edge fix) + 99 on one CU
    s.params.persistent = 0
    rows.append(s.run_iterations(37))                  # launch path (new handle) continues
    s.params.persistent = 4
    rows.append(s.run_iterations(64))
    s.params.persistent = 3
    rows.append(s.run_iterations(50))                  # small-N kernel
    s.params.persistent = 4
    rows.append(s.run_iterations(30))
    s.params.persistent = 0
    rows.append(s.run_iterations(1))
    rec = np.concatenate(rows, axis=0)
    o = orc.OracleSG(N, Re)
    want = oracle_rows(o, 282)
    assert rec.shape == (282, 8)
    assert np.max(np.abs(s.arrays.u.reshape(N + 1, N + 1) - o.u)) < 1e-12
    assert np.max(np.abs(s.arrays.p.reshape(N - 1, N - 1) - o.p)) < 1e-12
    assert rel(rec[:, 7], want[:, 7]) < 1e-12
    for c in range(1, 7):
        assert rel(rec[:, c], want[:, c]) < (1e-10 if c < 5 else 1e-9), c
    s.close()


def test_batches_pick_the_kernel_by_their_size():
    """persistent=-1 (the default): a batch of small trials is advanced by the small-N kernel (every trial on an XCD of its
    own) up to LDC_CU_AUTO_TRIALS(_T3, _M33) - 1 trials and by the trial-per-CU kernel from there on; both give the oracle's
    trajectory."""
    from solvers.spectral import ldc_lib as L
    from solvers.spectral.batched import BatchedSGSolver
    for N, B, want in ((16, L.CU_AUTO_TRIALS - 1, 3), (16, L.CU_AUTO_TRIALS, 4), (40, L.CU_AUTO_TRIALS_T3 - 1, 3),
                       (40, L.CU_AUTO_TRIALS_T3, 4), (32, L.CU_AUTO_TRIALS_M33 - 1, 3), (32, L.CU_AUTO_TRIALS_M33, 4),
                       (48, L.CU_AUTO_TRIALS, 3), (96, 4, 0)):
        trials = [dict(name="spectral", Re=100.0 + q, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=0.0,
                       max_iterations=10**9, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                       corner_treatment="smoothing", corner_smoothing=0.15, multigrid="none", check_every=64, graph_iters=16)
                  for q in range(B)]
        b = BatchedSGSolver(trials)
        K = 30
        recs = b.run_iterations(K)
        assert L.lib().ldc_batch_mode(b._batch) == want, (N, B)
        q = B - 1
        o = orc.OracleSG(N, trials[q]["Re"])
        ref = oracle_rows(o, K)
        assert np.max(np.abs(b.solvers[q].arrays.u.reshape(N + 1, N + 1) - o.u)) < 1e-12
        assert rel(recs[q][:, 7], ref[:, 7]) < 1e-12 and rel(recs[q][:, 4], ref[:, 4]) < 1e-10
        b.close()


def test_cu_batch_with_diverging_trials_latches_each_one_where_it_blows_up():
    """N = 16 at Re = 1000 blows up after ~2 300 iterations (the reference's scheme does: the oracle too).  In a batch large
    enough for the trial-per-CU kernel every work-group closes its own iterations: the NaN latch fires per trial, at the
    iteration the same trial alone reaches it, while the stable trials of the batch run on."""
    from solvers.spectral import ldc_lib as L
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    B = L.CU_AUTO_TRIALS + 2
    kw = dict(name="spectral", lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=16, ny=16, tolerance=1e-12, max_iterations=6000,
              basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing", multigrid="none",
              check_every=1024, graph_iters=16, nan_guard=True)
    trials = [dict(kw, Re=1000.0 if q % 2 else 100.0, corner_smoothing=0.02 + 0.001 * q) for q in range(B)]
    b = BatchedSGSolver(trials)
    ms = b.solve()
    assert L.lib().ldc_batch_mode(b._batch) == 4
    its = [m.iterations for m in ms]
    assert all(its[q] == 6000 for q in range(0, B, 2))                       # the stable ones reach the cap
    assert all(500 < its[q] < 6000 and not ms[q].converged for q in range(1, B, 2))
    for q in (1, B - 1 if (B - 1) % 2 else B - 2):
        one = SGSolver(**dict(trials[q], persistent=4))
        one.solve()
        assert one.metrics.iterations == its[q], q
        one.close()
    b.close()


def test_every_small_size_through_both_small_kernels():
    """N = 4 ... 43, every size once through the trial-per-CU kernel and once through the small-N kernel (12 iterations with
    E/Z/P against the oracle): no size between the parametrised ones hides an indexing case of its own."""
    from solvers.spectral import ldc_lib as L
    K = 12
    for N in range(4, 44):
        o = orc.OracleSG(N, 100.0)
        want = oracle_rows(o, K)
        for mode in (4, 3):
            s = make(N, 100.0, persistent=mode, check_every=64)
            rec = s.run_iterations(K)
            assert L.lib().ldc_solver_mode(s._handle) == mode, (N, mode)
            M = N + 1
            assert np.max(np.abs(s.arrays.u.reshape(M, M) - o.u)) < 1e-12, (N, mode)
            assert np.max(np.abs(s.arrays.p.reshape(M - 2, M - 2) - o.p)) < 1e-12, (N, mode)
            assert rel(rec[:, 7], want[:, 7]) < 1e-12, (N, mode)
            for c in range(1, 7):
                assert rel(rec[:, c], want[:, c]) < (1e-10 if c < 5 else 1e-9), (N, mode, c)
            s.close()


def test_cu_variants_and_legendre_vs_reference(golden_dir):
    """The reference's short runs with non-default parameters (Saad lid, other CFL / beta^2, Lx != Ly with another lid speed,
    odd N: g4b) and with the Legendre basis (g12: both EDGE sizes, 17 and 33 nodes) through the trial-per-CU kernel."""
    g = np.load(golden_dir / "g4b_variants.npz")
    meta = json.loads((golden_dir / "g4b_variants.json").read_text())
    for name, c in meta.items():
        s = make(c["N"], c["Re"], **c["kw"])
        rec = s.run_iterations(c["K"])
        assert mode_of(s) == 4, name
        for k, a in (("u", s.arrays.u), ("v", s.arrays.v), ("p", s.arrays.p)):
            assert np.max(np.abs(a - g[f"{name}_{k}"])) < 1e-12, (name, k)
        assert rel(rec[:, 7], g[f"{name}_dt"]) < 1e-12, name
        assert rel(rec[:, 1:4], g[f"{name}_res"]) < 1e-10, name
        assert rel(rec[:, 6], g[f"{name}_P"]) < 1e-9, name
        s.close()
    g = np.load(golden_dir / "g12_legendre.npz")
    meta = json.loads((golden_dir / "g12_legendre.json").read_text())
    for tag, c in meta.items():
        s = make(c["N"], c["Re"], basis_type="legendre")
        rec = s.run_iterations(c["K"])
        assert mode_of(s) == 4, tag
        for k, a in (("u", s.arrays.u), ("v", s.arrays.v), ("p", s.arrays.p)):
            assert np.max(np.abs(a - g[f"{tag}_{k}"])) < 1e-11, (tag, k)
        assert rel(rec[:, 7], g[f"{tag}_dt"]) < 1e-12 and rel(rec[:, 1:4], g[f"{tag}_res"]) < 1e-10
        for col, k in ((4, "E"), (5, "Z"), (6, "P")):
            assert rel(rec[:, col], g[f"{tag}_{k}"]) < 1e-9, (tag, k)
        s.close()
