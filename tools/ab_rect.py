"""nx != ny: microseconds per full iteration (E / Z / P) of ONE trial on the launch-per-stage path (persistent=0) against the
library's own choice (one-XCD kernel up to M = 80, chip-wide kernel above).
    python tools/ab_rect.py            (development aid; log: profiles/r04_rect_ab.log)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402


def us_per_iteration(nx, ny, mode, K=4096):
    s = SGSolver(name="spectral", Re=100.0, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=nx, ny=ny, tolerance=0.0, max_iterations=10**9,
                 basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
                 multigrid="none", check_every=K, graph_iters=64, persistent=mode)
    s.run_iterations(256)
    best = 1e30
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.run_iterations(K)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / K * 1e6)
    m = int(L.lib().ldc_solver_mode(s._handle))
    s.close()
    return best, m


for nx, ny in [(24, 40), (64, 32), (32, 64), (30, 100), (48, 129), (129, 48), (64, 200), (128, 255)]:
    a, ma = us_per_iteration(nx, ny, 0)
    b, mb = us_per_iteration(nx, ny, -1)
    print(f"nx={nx:4d} ny={ny:4d}: launch path (mode {ma}) {a:7.2f} us/iteration | library's choice (mode {mb}) {b:7.2f}", flush=True)
