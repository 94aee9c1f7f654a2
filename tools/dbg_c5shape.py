"""Bisect helper for the config-5-shape test (two FSG batches on two streams): python tools/dbg_c5shape.py <variant>
variants: threads_auto | threads_launch | one_batch | sequential   (development aid)"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import faulthandler; faulthandler.enable(all_threads=True)
import numpy as np
from solvers.spectral.batched import BatchedFSGSolver, solve_concurrently
v = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cs = [0.02 + 0.011 * q for q in range(8)]
base = dict(name="spectral_fsg", Re=1000.0, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=1e-6,
            max_iterations=150, basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing",
            multigrid="fsg", n_levels=2, coarse_tolerance_factor=1.0, prolongation_method="fft",
            restriction_method="fft", check_every=64, graph_iters=16, persistent=(0 if v == "threads_launch" else -1))
trials = [dict(base, corner_smoothing=c) for c in cs]
if v in ("threads_auto", "threads_launch"):
    halves = [BatchedFSGSolver(trials[:4]), BatchedFSGSolver(trials[4:])]
    solve_concurrently(halves)
    sol = halves[0].solvers + halves[1].solvers
elif v == "one_batch":
    b = BatchedFSGSolver(trials); b.solve(); sol = b.solvers
else:
    halves = [BatchedFSGSolver(trials[:4]), BatchedFSGSolver(trials[4:])]
    for h in halves: h.solve()
    sol = halves[0].solvers + halves[1].solvers
print(v, "ok", [s.metrics.iterations for s in sol], float(np.abs(sol[0].arrays.u).max()), flush=True)
