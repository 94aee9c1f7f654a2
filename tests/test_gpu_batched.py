"""Batched trials: B independent solves advanced by the same launches must be
bit-identical to the same solves run one after another."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def kw(N, Re, cs=0.15, **extra):
    d = dict(name="spectral", Re=float(Re), lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=1e-6,
             max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
             corner_treatment="smoothing", corner_smoothing=cs, multigrid="none", check_every=256, graph_iters=8)
    d.update(extra)
    return d


def batch_mode(b):
    from solvers.spectral import ldc_lib as L
    return int(L.lib().ldc_batch_mode(b._batch))


@pytest.mark.parametrize("persistent", [0, -1])
@pytest.mark.parametrize("N", [32, 24, 64])
def test_batched_iterations_equal_individual_runs(N, persistent):
    """persistent=0: the shared launches of the launch path (blockIdx.y = trial, batched graphs); -1: what a sweep gets by
    default at these sizes (the one-XCD kernel, mode 3).  Both against one-after-another runs in the SAME mode."""
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    trials = [kw(N, 100, 0.15), kw(N, 400, 0.10), kw(N, 50, 0.30, CFL=1.0), kw(N, 250, 0.05, corner_treatment="saad"),
              kw(N, 100, 0.15, beta_squared=3.0)]
    trials = [dict(t, persistent=persistent) for t in trials]
    b = BatchedSGSolver(trials)
    recs = b.run_iterations(150)
    assert batch_mode(b) == (0 if persistent == 0 else 3)
    for t, s, r in zip(trials, b.solvers, recs):
        one = SGSolver(**t)
        r1 = one.run_iterations(150)
        assert np.array_equal(r, r1)
        assert np.array_equal(s.arrays.u, one.arrays.u) and np.array_equal(s.arrays.p, one.arrays.p)
        one.close()
    b.close()


@pytest.mark.parametrize("N,Re,K", [(32, 100, 500), (64, 400, 1000)])
def test_batched_trial_vs_reference_fixture(golden_dir, N, Re, K):
    """A trial INSIDE a batch (other trials with other Re / lid profiles around it) against the reference's own
    K-step trajectory: state <= 1e-12, dt <= 1e-12 rel, norms / E / Z / P <= 1e-10 rel (the bar of
    tests/test_gpu_parity.py::test_trajectory_vs_reference)."""
    from solvers.spectral.batched import BatchedSGSolver
    g = np.load(golden_dir / f"g4_traj_N{N}_Re{Re}_K{K}.npz")
    rel = lambda a, b: np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)     # noqa: E731
    trials = [kw(N, 250, 0.05), kw(N, Re, 0.15), kw(N, 50, 0.30, CFL=1.0), kw(N, Re, 0.15)]
    b = BatchedSGSolver(trials)
    recs = b.run_iterations(K)
    for q in (1, 3):
        s, rec = b.solvers[q], recs[q]
        assert rec.shape == (K, 8)
        assert np.max(np.abs(s.arrays.u - g["u"])) < 1e-12
        assert np.max(np.abs(s.arrays.v - g["v"])) < 1e-12
        assert np.max(np.abs(s.arrays.p - g["p"])) < 1e-12
        assert rel(rec[:, 7], g["dt"]) < 1e-12
        assert rel(rec[:, 1:4], g["res"]) < 1e-10
        for col, key in ((4, "E"), (5, "Z"), (6, "P")):
            assert rel(rec[:, col], g[key]) < 1e-10, key
    assert not np.array_equal(b.solvers[0].arrays.u, b.solvers[1].arrays.u)      # the neighbours are different flows
    b.close()


def test_batched_trials_keep_their_own_iteration_caps():
    """One process per trial in the reference: a sweep over solver.max_iterations gives every trial ITS cap.
    In a batch the host latches a trial when it reaches its cap (code 3); state and history equal the
    stand-alone capped solves, and the wall-time shares add up to the batch's wall time."""
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    trials = [kw(32, 100, max_iterations=150), kw(32, 100, max_iterations=700), kw(32, 400, max_iterations=333),
              kw(32, 100, tolerance=1e-2, max_iterations=5000)]
    b = BatchedSGSolver(trials)
    ms = b.solve()
    assert [m.iterations for m in ms[:3]] == [150, 700, 333] and not any(m.converged for m in ms[:3])
    assert ms[3].converged and ms[3].iterations < 5000
    for t, s, m in zip(trials, b.solvers, ms):
        one = SGSolver(**t)
        one.solve()
        assert (m.iterations, m.converged) == (one.metrics.iterations, one.metrics.converged)
        assert np.array_equal(s.fields.u, one.fields.u) and np.array_equal(s.fields.p, one.fields.p)
        assert s.time_series.energy == one.time_series.energy
        one.close()
    assert sum(m.wall_time_seconds for m in ms) == pytest.approx(b.batch_seconds, rel=1e-9)
    with pytest.raises(ValueError, match="diagnostics"):
        BatchedSGSolver([kw(16, 100), kw(16, 100, diagnostics=False)]).solve()
    b.close()


def test_batched_solve_latches_each_trial_independently():
    """Three trials converge at different iterations; a fourth (N=16, Re=400 diverges with CFL 1.5,
    SURVEY section 5) leaves through the NaN latch.  All equal their stand-alone solves."""
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    trials = [kw(16, 100, tolerance=1e-3), kw(16, 150, tolerance=1e-4), kw(16, 100, tolerance=3e-4, corner_smoothing=0.3),
              kw(16, 400, tolerance=1e-6, nan_guard=True)]
    b = BatchedSGSolver(trials)
    ms = b.solve(max_iter=20000)
    its = []
    for t, s, m in zip(trials, b.solvers, ms):
        one = SGSolver(**t)
        one.solve(max_iter=20000)
        assert m.converged == one.metrics.converged
        assert m.iterations == one.metrics.iterations
        assert np.array_equal(s.fields.u, one.fields.u, equal_nan=True)
        if m.converged:
            assert m.psi_min == one.metrics.psi_min and m.final_palinstrophy == one.metrics.final_palinstrophy
            assert s.time_series.rel_iter_residual == one.time_series.rel_iter_residual
        its.append(m.iterations)
        one.close()
    assert [m.converged for m in ms] == [True, True, True, False]
    assert len(set(its)) == 4 and its[3] < 20000          # all stopped at different iterations
    b.close()


def test_batch_rejects_mixed_sizes():
    from solvers.spectral.batched import BatchedSGSolver
    with pytest.raises(ValueError):
        BatchedSGSolver([kw(16, 100), kw(32, 100)])


def fsg_kw(N, Re, cs=0.15, **extra):
    d = kw(N, Re, cs, name="spectral_fsg", multigrid="fsg", n_levels=2, coarse_tolerance_factor=1.0,
           prolongation_method="fft", restriction_method="fft", max_iterations=200000)
    d.update(extra)
    return d


def test_batched_fsg_equals_individual_fsg_solves():
    """Level by level the trials' smoothers share launches; every trial keeps its own latch per level.
    Three-level hierarchy, different Re / lid regularisation / tolerances; one trial is capped by
    max_iterations on every level.  Results equal the stand-alone FSG solves bit for bit."""
    from solvers.spectral.batched import BatchedFSGSolver
    from solvers.spectral.fsg import FSGSolver
    trials = [fsg_kw(48, 100, 0.15, n_levels=3, tolerance=1e-4), fsg_kw(48, 200, 0.10, n_levels=3, tolerance=3e-4),
              fsg_kw(48, 100, 0.30, n_levels=3, tolerance=1e-4, coarse_tolerance_factor=2.0),
              fsg_kw(48, 50, 0.15, n_levels=3, tolerance=1e-5, CFL=1.0)]
    b = BatchedFSGSolver(trials)
    assert b.orders == [12, 24, 48]
    ms = b.solve()
    for t, s, m in zip(trials, b.solvers, ms):
        one = FSGSolver(**t)
        one.solve()
        assert (m.converged, m.iterations) == (one.metrics.converged, one.metrics.iterations)
        assert np.array_equal(s.fields.u, one.fields.u) and np.array_equal(s.fields.p, one.fields.p)
        assert m.psi_min == one.metrics.psi_min and m.final_palinstrophy == one.metrics.final_palinstrophy
        one.close()
    assert all(m.converged for m in ms) and len({m.iterations for m in ms}) == 4
    b.close()


def test_batch_workspace_is_not_filled_behind_the_librarys_back():
    """ldc_batch_create copies the argument blocks into the caller's workspace on a stream of its own.  The workspace used to
    be a torch.zeros tensor: behind a long launch of another worker thread the fill kernel ran AFTER those copies, wiped the
    blocks, and the batch's kernels read null pointers (a memory access fault in a 600-trial search round).  Since ABI 7 the
    library itself makes its stream wait for the caller's (an event) before it copies -- checked without provoking the fault: a
    long sleep AND a fill of the workspace-to-be are queued in front on this stream, and when the batch exists the stream is
    idle and the blocks are intact (the batch runs)."""
    import torch
    from solvers.spectral.batched import BatchedSGSolver
    b = BatchedSGSolver([kw(32, 100.0), kw(32, 200.0)])
    b._alloc_workspace = lambda nbytes, dev: torch.zeros(nbytes, dtype=torch.uint8, device=dev)     # the fill of round 3
    torch.cuda.synchronize()
    torch.cuda._sleep(200_000_000)                    # ~0.1 s of an idle kernel on this stream, the fill queues behind it
    assert not torch.cuda.current_stream().query()
    b._ensure_batch([0.0, 0.0])
    assert torch.cuda.current_stream().query()        # the library waited for this stream before it copied, and for its copies
    assert int(torch.count_nonzero(b._ws)) > 0        # ... so the fill did not come last: the argument blocks are there
    rec = b.run_iterations(40)
    assert np.all(np.isfinite(rec[0])) and np.all(np.isfinite(rec[1]))
    b.close()


def test_kernel_attributes_are_set_once_per_device_not_per_create():
    """hipFuncSetAttribute (dynamic LDS above 64 KiB) once per device and process, before the first handle exists -- never
    again at later creates, which may run in one host thread while another launches the same kernels (the inferred cause of
    round 3's two silent aborts).  The library counts the rounds."""
    from solvers.spectral import ldc_lib as L
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    one = SGSolver(**kw(24, 100.0))
    one.run_iterations(4)
    assert L.lib().ldc_attribute_rounds() == 1
    for n in (16, 32, 48, 96):
        s = SGSolver(**kw(n, 100.0))
        s.run_iterations(4)
        s.close()
    b = BatchedSGSolver([kw(32, 100.0), kw(32, 200.0)])
    b.run_iterations(8)
    b.close()
    one.close()
    assert L.lib().ldc_attribute_rounds() == 1


def test_batched_fsg_where_every_trial_diverges_on_the_coarse_level():
    """The reference's Optuna experiment samples N = 30 at Re = 1000 (conf/experiment/optimization/corner_smoothing.yaml): its
    coarse level N = 15 blows up within a few thousand iterations, the NaN latch ends the trial ("early NaN/Inf detection
    exits diverging runs quickly") and nothing is prolongated.  A batch whose trials ALL end that way must finish like the
    stand-alone solves -- same iteration counts, converged = False -- instead of building an empty batch for the next level
    (found by running that experiment file as it stands: tools/sweep_report.py optuna_ref)."""
    from solvers.spectral.batched import BatchedFSGSolver
    from solvers.spectral.fsg import FSGSolver
    trials = [fsg_kw(30, 1000, 0.03, tolerance=1e-6, max_iterations=500000),
              fsg_kw(30, 1000, 0.08, tolerance=1e-6, max_iterations=500000)]
    b = BatchedFSGSolver(trials)
    assert b.orders == [15, 30]
    ms = b.solve()
    for t, m in zip(trials, ms):
        one = FSGSolver(**t)
        one.solve()
        assert not m.converged and not one.metrics.converged
        assert m.iterations == one.metrics.iterations and 0 < m.iterations < 50000
        one.close()
    b.close()


def test_batched_fsg_trial_vs_reference_fixture(golden_dir):
    """A batched FSG trial against the reference's own capped two-level run (g8: cap300_N32_Re100)."""
    import json
    from solvers.spectral.batched import BatchedFSGSolver
    g = np.load(golden_dir / "g8_fsg_runs.npz")
    c = json.loads((golden_dir / "g8_fsg_runs.json").read_text())["cap300_N32_Re100"]
    base = fsg_kw(32, 100.0, 0.15, tolerance=1e-6, max_iterations=300)
    b = BatchedFSGSolver([fsg_kw(32, 400.0, 0.10, max_iterations=300), base, fsg_kw(32, 50.0, 0.3, max_iterations=300)])
    ms = b.solve()
    s, m, ref = b.solvers[1], ms[1], c["metrics"]
    assert m.iterations == ref["iterations"] and m.converged == ref["converged"]
    for f in ("u", "v", "p"):
        assert np.max(np.abs(getattr(s.arrays, f) - g[f"cap300_N32_Re100_{f}"])) < 1e-10, f
    for key in ("u_momentum_residual", "continuity_residual", "final_energy", "final_enstrophy", "psi_min"):
        assert getattr(m, key) == pytest.approx(ref[key], rel=1e-7, abs=1e-9), key
    b.close()


def test_batched_fsg_rejects_mixed_hierarchies():
    from solvers.spectral.batched import BatchedFSGSolver
    with pytest.raises(ValueError):
        BatchedFSGSolver([fsg_kw(32, 100), fsg_kw(32, 100, n_levels=1)])


@pytest.mark.parametrize("persistent", [0, -1])
def test_two_halves_on_two_streams_equal_stand_alone_solves(persistent):
    """main.py runs a rank's equal-N trials as two batches on two HIP streams (solve_concurrently): same kernels per
    trial, so converged solves keep their iteration counts and fields bit for bit; the wall-time shares add up."""
    from solvers.spectral.batched import BatchedSGSolver, solve_concurrently
    from solvers.spectral.sg import SGSolver
    trials = [kw(32, 100, 0.15, tolerance=1e-4), kw(32, 400, 0.10, tolerance=1e-4), kw(32, 50, 0.30, tolerance=1e-4),
              kw(32, 250, 0.05, tolerance=1e-4, max_iterations=700), kw(32, 100, 0.15, tolerance=1e-4, beta_squared=3.0)]
    trials = [dict(t, persistent=persistent) for t in trials]
    halves = [BatchedSGSolver(trials[:3]), BatchedSGSolver(trials[3:])]
    wall = solve_concurrently(halves)
    assert all(batch_mode(b) == (0 if persistent == 0 else 3) for b in halves)
    solvers = halves[0].solvers + halves[1].solvers
    assert halves[0].batch_seconds == halves[1].batch_seconds == wall and halves[0].batch_size == 5
    assert abs(sum(s.metrics.wall_time_seconds for s in solvers) - wall) < 1e-6 * max(1.0, wall)
    for t, s in zip(trials, solvers):
        one = SGSolver(**t)
        one.solve()
        assert s.metrics.iterations == one.metrics.iterations and s.metrics.converged == one.metrics.converged
        assert np.array_equal(s.fields.u, one.fields.u) and np.array_equal(s.fields.p, one.fields.p)
        one.close()
    assert solvers[3].metrics.iterations == 700 and not solvers[3].metrics.converged
    for b in halves:
        b.close()


@pytest.mark.parametrize("persistent", [0, -1])
def test_fsg_batches_on_two_streams_equal_stand_alone_solves(persistent):
    from solvers.spectral.batched import BatchedFSGSolver, solve_concurrently
    from solvers.spectral.fsg import FSGSolver
    base = dict(name="spectral_fsg", nx=32, ny=32, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                corner_treatment="smoothing", multigrid="fsg", n_levels=2, coarse_tolerance_factor=10.0, tolerance=1e-5,
                max_iterations=4000, check_every=128, graph_iters=16, persistent=persistent)
    trials = [dict(base, Re=100.0, corner_smoothing=0.15), dict(base, Re=400.0, corner_smoothing=0.08),
              dict(base, Re=200.0, corner_smoothing=0.25)]
    halves = [BatchedFSGSolver(trials[:2]), BatchedFSGSolver(trials[2:])]
    solve_concurrently(halves)
    for t, s in zip(trials, halves[0].solvers + halves[1].solvers):
        one = FSGSolver(**t)
        one.solve()
        assert s.metrics.iterations == one.metrics.iterations
        assert np.array_equal(s.fields.u, one.fields.u) and np.array_equal(s.fields.v, one.fields.v)
        one.close()
    for b in halves:
        b.close()


def test_worker_streams_come_in_three_priorities():
    """The library hands out non-blocking streams of every priority level the hardware has (three here; torch offers
    two); run_concurrently wraps them as torch streams, one per worker, and keeps them."""
    import ctypes as C
    import torch
    from solvers.spectral import ldc_lib as L
    from solvers.spectral import batched
    lo, hi = C.c_int(), C.c_int()
    assert L.lib().ldc_stream_priority_range(C.byref(lo), C.byref(hi)) == 0
    assert lo.value > 0 > hi.value                         # least = 1, greatest = -1 on MI355X
    assert L.lib().ldc_stream_create(0, None) == -1        # LDC_E_ARG
    for prio in (lo.value, 0, hi.value):
        h = C.c_void_p()
        assert L.lib().ldc_stream_create(prio, C.byref(h)) == 0 and h.value
        s = torch.cuda.ExternalStream(h.value)
        with torch.cuda.stream(s):
            x = torch.ones(1024, device="cuda").sum()
        s.synchronize()
        assert float(x) == 1024.0
        assert L.lib().ldc_stream_destroy(h) == 0
    seen = []
    batched.run_concurrently([0, 1, 2], lambda k: seen.append((k, torch.cuda.current_stream().cuda_stream)))
    assert len({ptr for _, ptr in seen}) == 3 and sorted(k for k, _ in seen) == [0, 1, 2]
    again = []
    batched.run_concurrently([0, 1, 2], lambda k: again.append((k, torch.cuda.current_stream().cuda_stream)))
    assert dict(seen) == dict(again)                        # the same three streams every time


def test_batches_built_and_closed_while_another_thread_replays_graphs():
    """The library's hot calls (hipGraphLaunch on the caller's stream) take no lock, while graph capture, instantiation
    and destruction of graph executables run under its process-wide mutex on a kept private stream.  Thread A builds,
    runs briefly and closes batch after batch (captures + instantiations + exec destructions) while thread B keeps
    replaying the graphs of a batch of its own and waiting on its stream: B's records must equal an undisturbed run
    bit for bit, nothing may abort (round 2's runtime abort with per-handle capture streams: DESIGN.md 3, streams)."""
    import threading
    import torch
    from solvers.spectral.batched import BatchedSGSolver, run_concurrently
    # persistent=0: the launch path, the only one that captures, instantiates and replays hipGraphs (in auto mode these sizes
    # take the one-XCD kernel, whose launches ldc_lib.resident_lock serialises -- nothing of this test would be exercised)
    trials_b = [kw(32, 100, 0.15, graph_iters=8, persistent=0), kw(32, 400, 0.10, graph_iters=8, persistent=0)]
    ref = BatchedSGSolver(trials_b)
    want = ref.run_iterations(1200, diagnostics=False)
    assert batch_mode(ref) == 0
    ref.close()
    got, built, modes = {}, [], []

    def job(which):
        if which == "A":
            for k in range(6):
                b = BatchedSGSolver([kw(24, 100 + 50 * k, 0.1, graph_iters=4, persistent=0),
                                     kw(24, 200, 0.2, graph_iters=4, persistent=0)])
                b.run_iterations(16, diagnostics=(k % 2 == 0))       # capture + instantiate (both graph flavours in turn)
                modes.append(batch_mode(b))
                b.close()                                            # hipGraphExecDestroy under the mutex
                built.append(k)
        else:
            b = BatchedSGSolver(trials_b)
            rows = [b.run_iterations(100, diagnostics=False) for _ in range(12)]      # replays + stream-level waits
            modes.append(batch_mode(b))
            got["B"] = [np.concatenate([r[q] for r in rows], axis=0) for q in range(2)]
            b.close()

    run_concurrently(["A", "B"], job)
    assert built == list(range(6)) and modes == [0] * 7
    for w, g in zip(want, got["B"]):
        assert np.array_equal(w, g)
