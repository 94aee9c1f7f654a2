"""A/B of the chip-wide trial kernel (persistent=5) against the launch path (persistent=0): microseconds per iteration of one
trial (SG step-only, SG with E/Z/P, smoother mode).
    python tools/ab_wide.py [N ...]          (default 128 255)
(development aid; the committed logs are profiles/r04_wide_*.log)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [128, 256]
K = int(os.environ.get("AB_K", "2048"))
MODES = [int(x) for x in os.environ.get("AB_MODES", "0,5").split(",")]


def kw(N, mode, **extra):
    d = dict(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5,
             tolerance=0.0, max_iterations=10**9, check_every=K, graph_iters=64, persistent=mode)
    d.update(extra)
    return d


def timed(fn, reps=3):
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


for N in sizes:
    for label, diag, smoother in (("SG step-only", False, False), ("SG with E/Z/P", True, False), ("smoother", False, True)):
        row = []
        for mode in MODES:
            s = SGSolver(**kw(N, mode))
            if smoother:
                s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
            s.run_iterations(256, diagnostics=diag)
            got = L.lib().ldc_solver_mode(s._handle)
            dt = timed(lambda: s.run_iterations(K, diagnostics=diag))
            row.append((got, dt / K * 1e6))
            s.close()
        print(f"N={N:3d} {label:14s}: " + "   ".join(f"mode {m}: {us:7.2f} us/iter" for m, us in row), flush=True)
