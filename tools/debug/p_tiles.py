"""Which tiles' palinstrophy partials differ between two identical runs of the chip-wide kernel (a race shows as a difference)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np
from solvers.spectral.sg import SGSolver
kw = dict(name="spectral", Re=1000.0, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=256, ny=256, tolerance=1e-6,
          max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing",
          corner_smoothing=0.15, multigrid="none", check_every=512, graph_iters=16, persistent=5)
A, B = SGSolver(**kw), SGSolver(**kw)
T = 16
seen = {}
for rep in range(80):
    A.run_iterations(5); B.run_iterations(5)
    pa, pb = A.d["partials"].cpu().numpy(), B.d["partials"].cpu().numpy()
    st = A._part_stride
    for slab, name in ((1, "Z0"), (2, "Z1"), (3, "P0"), (4, "P1")):
        x, y = pa[slab * st: slab * st + T * T], pb[slab * st: slab * st + T * T]
        bad = np.nonzero(x != y)[0]
        for bx in bad:
            seen.setdefault((name[0], int(bx) // T, int(bx) % T), []).append(abs(x[bx] - y[bx]) / max(abs(x[bx]), 1e-300))
    ua, ub = A.d["U"].cpu().numpy(), B.d["U"].cpu().numpy()
    if not np.array_equal(ua, ub):
        print("rep", rep, "STATE differs", np.abs(ua - ub).max())
for k in sorted(seen):
    print(k, len(seen[k]), "max rel", max(seen[k]))
print("done", len(seen))
