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


@pytest.mark.parametrize("N", [32, 24, 64])
def test_batched_iterations_equal_individual_runs(N):
    from solvers.spectral.batched import BatchedSGSolver
    from solvers.spectral.sg import SGSolver
    trials = [kw(N, 100, 0.15), kw(N, 400, 0.10), kw(N, 50, 0.30, CFL=1.0), kw(N, 250, 0.05, corner_treatment="saad"),
              kw(N, 100, 0.15, beta_squared=3.0)]
    b = BatchedSGSolver(trials)
    recs = b.run_iterations(150)
    for t, s, r in zip(trials, b.solvers, recs):
        one = SGSolver(**t)
        r1 = one.run_iterations(150)
        assert np.array_equal(r, r1)
        assert np.array_equal(s.arrays.u, one.arrays.u) and np.array_equal(s.arrays.p, one.arrays.p)
        one.close()
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
