"""Which rows of the palinstrophy column differ from the reference's run (N=256, K=200, tail layout), several repeats."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np
from solvers.spectral.sg import SGSolver
g = np.load("tests/golden/g4c_traj_N256_Re1000_K200.npz")
for rep in range(4):
    s = SGSolver(name="spectral", Re=1000.0, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=256, ny=256, tolerance=1e-6,
                 max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing",
                 corner_smoothing=0.15, multigrid="none", check_every=512, graph_iters=16, persistent=5)
    rec = s.run_iterations(200)
    for col, key in ((4, "E"), (5, "Z"), (6, "P")):
        d = np.abs(rec[:, col] - g[key]) / np.abs(g[key])
        bad = np.nonzero(d > 1e-10)[0]
        print(rep, key, "max rel", d.max(), "bad rows", bad[:20], d[bad[:6]])
    s.close()
