"""nx != ny: the reference builds the x and y grids independently (sg.py:103-119).  On the device the arrays, the tiling
and LD are built for M = max(nx, ny) + 1, the shorter axis is zero padding, the kernels take wall / lid / interior from
(Mx, My); the launch-per-stage path, the one-XCD kernel (M <= 80) and the chip-wide kernel (above) run them (include/ldc_hip.h, ldc_problem::Mx).  Pinned
against the reference's own runs (tests/golden/g13_unequal_grids.*, made by tests/golden/make_golden.py G13) and against the oracle."""
import json

import numpy as np
import pytest

from oracle import ldc_oracle as orc
from test_gpu_xcd import rel


def _load(golden_dir):
    return np.load(golden_dir / "g13_unequal_grids.npz"), json.loads((golden_dir / "g13_unequal_grids.json").read_text())


def make(nx, ny, Re, **kw):
    from solvers.spectral.sg import SGSolver
    args = dict(name="spectral", Re=float(Re), lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=nx, ny=ny,
                tolerance=1e-6, max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5,
                beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
                multigrid="none", check_every=256, graph_iters=16)
    args.update(kw)
    return SGSolver(**args)


@pytest.mark.parametrize("name", ["nx24_ny40", "nx40_ny20", "nx17_ny32"])
def test_oracle_matches_the_reference_on_unequal_grids(golden_dir, name):
    g, meta = _load(golden_dir)
    c = meta[name]
    o = orc.OracleSG(c["nx"], c["Re"], ny=c["ny"], **c["kw"])
    dts = [o.step() for _ in range(c["K"])]
    Mx, My = c["nx"] + 1, c["ny"] + 1
    assert np.max(np.abs(o.u - g[f"{name}_u"].reshape(Mx, My))) < 1e-13
    assert np.max(np.abs(o.v - g[f"{name}_v"].reshape(Mx, My))) < 1e-13
    assert np.max(np.abs(o.p - g[f"{name}_p"].reshape(Mx - 2, My - 2))) < 1e-13
    assert rel(np.array(dts), g[f"{name}_dt"]) < 1e-13
    assert np.max(np.abs(o.vorticity() - g[f"{name}_omega"].reshape(Mx, My))) < 1e-12
    assert np.max(np.abs(o.streamfunction() - g[f"{name}_psi"])) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("name", ["nx24_ny40", "nx40_ny20", "nx17_ny32"])
def test_gpu_trajectory_vs_reference_on_unequal_grids(golden_dir, name, mode):
    """K steps from rest with every record column, omega, psi and the vortex metrics of the end state: x longer than y, y
    longer than x, Lx != Ly, the Saad lid; one of the sizes has 16 T + 1 nodes on one axis only (no tail layout).  On the
    launch-per-stage path and on the one-XCD kernel (the library's own choice for a lone trial of these sizes)."""
    from solvers.spectral import ldc_lib as L
    g, meta = _load(golden_dir)
    c = meta[name]
    s = make(c["nx"], c["ny"], c["Re"], persistent=mode, **c["kw"])
    rec = s.run_iterations(c["K"])
    assert L.lib().ldc_solver_mode(s._handle) == mode
    assert rec.shape == (c["K"], 8)
    assert np.max(np.abs(s.arrays.u - g[f"{name}_u"])) < 1e-12
    assert np.max(np.abs(s.arrays.v - g[f"{name}_v"])) < 1e-12
    assert np.max(np.abs(s.arrays.p - g[f"{name}_p"])) < 1e-12
    assert rel(rec[:, 7], g[f"{name}_dt"]) < 1e-12
    assert np.max(np.abs(rec[:, 0] - g[f"{name}_rel"]) / (np.abs(g[f"{name}_rel"]) + 1e-9)) < 1e-8
    assert rel(rec[:, 1:4], g[f"{name}_res"]) < 1e-10
    assert rel(rec[:, 4], g[f"{name}_E"]) < 1e-10
    assert rel(rec[:, 5], g[f"{name}_Z"]) < 1e-10
    assert rel(rec[:, 6], g[f"{name}_P"]) < 1e-10
    assert rel(s._compute_vorticity(), g[f"{name}_omega"]) < 1e-10
    psi, _, _ = s._compute_streamfunction()
    assert psi.shape == g[f"{name}_psi"].shape
    assert np.max(np.abs(psi - g[f"{name}_psi"])) < 1e-10 * np.max(np.abs(g[f"{name}_psi"]))
    vm = s.compute_vortex_metrics()
    for k, v in c["vortex"].items():
        assert vm[k] == pytest.approx(v, rel=1e-8, abs=1e-10), k
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("auto", [True, False])
@pytest.mark.parametrize("nx,ny,Re", [(64, 32, 400.0), (30, 100, 100.0), (48, 129, 100.0)])
def test_gpu_records_vs_oracle_on_unequal_grids(nx, ny, Re, auto):
    """Larger and more lopsided grids against the oracle (several tiles per axis, one axis with 16 T + 2 nodes), on the
    library's own choice -- the one-XCD kernel up to M = 80, the chip-wide kernel above (index M-1 inside the tiles) -- and on
    the launch-per-stage path."""
    K = 40
    o = orc.OracleSG(nx, Re, ny=ny)
    up, rows = None, []
    for _ in range(K):
        up, vp = o.u.copy(), o.v.copy()
        dt = o.step()
        nrm = lambda a, b: np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12)      # noqa: E731
        rows.append([max(nrm(o.u, up), nrm(o.v, vp)), *o.residual_norms(), o.energy(), o.enstrophy(), o.palinstrophy(), dt])
    want = np.array(rows)
    s = make(nx, ny, Re) if auto else make(nx, ny, Re, persistent=0)
    rec = s.run_iterations(K)
    from solvers.spectral import ldc_lib as L
    assert L.lib().ldc_solver_mode(s._handle) == ((3 if max(nx, ny) <= 79 else 5) if auto else 0)     # (64 x 32: 5 x 5 tiles on one XCD)
    Mx, My = nx + 1, ny + 1
    assert np.max(np.abs(s.arrays.u.reshape(Mx, My) - o.u)) < 1e-12
    assert np.max(np.abs(s.arrays.v.reshape(Mx, My) - o.v)) < 1e-12
    assert np.max(np.abs(s.arrays.p.reshape(Mx - 2, My - 2) - o.p)) < 1e-12
    assert rel(rec[:, 7], want[:, 7]) < 1e-12
    for col in range(1, 7):
        assert rel(rec[:, col], want[:, col]) < (1e-10 if col < 5 else 1e-9), col
    s.close()


@pytest.mark.gpu
def test_gpu_solve_and_post_processing_on_unequal_grids():
    """A whole solve() with nx != ny: fields come back on the (nx+1) x (ny+1) grid, metrics and validation run, and the
    iteration count equals the oracle's."""
    s = make(20, 28, 100.0, tolerance=1e-4, check_every=128)
    s.solve()
    o = orc.OracleSG(20, 100.0, ny=28)
    its, conv, _ = o.solve(tolerance=1e-4, diagnostics=False)
    assert s.metrics.converged and conv and s.metrics.iterations == its
    assert s.fields.u.size == 21 * 29 and s.fields.p.size == 21 * 29
    assert np.max(np.abs(s.fields.u.reshape(21, 29) - o.u)) < 1e-10
    assert np.isfinite(s.metrics.psi_min) and s.metrics.psi_min < 0
    assert set(s.ghia_error()) >= {"u_rms", "v_rms"}
    s.close()


@pytest.mark.gpu
def test_fsg_and_batches_of_unlike_grids_are_refused():
    from solvers.spectral.batched import BatchedSGSolver
    from test_fsg import make_fsg
    with pytest.raises(NotImplementedError):
        make_fsg(32, 100.0, ny=24)
    kw = dict(name="spectral", Re=100.0, lid_velocity=1.0, Lx=1.0, Ly=1.0, tolerance=1e-6, max_iterations=1000,
              basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
              multigrid="none")
    with pytest.raises(ValueError):
        BatchedSGSolver([dict(kw, nx=24, ny=40), dict(kw, nx=40, ny=24)])
    b = BatchedSGSolver([dict(kw, nx=24, ny=40), dict(kw, nx=24, ny=40, Re=200.0)])      # equal unequal grids share launches
    recs = b.run_iterations(30)
    o = orc.OracleSG(24, 200.0, ny=40)
    for _ in range(30):
        o.step()
    assert np.max(np.abs(b.solvers[1].arrays.u.reshape(25, 41) - o.u)) < 1e-12 and np.all(np.isfinite(recs[0]))
    b.close()
