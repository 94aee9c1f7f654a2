"""A/B of the small-N trial kernel (persistent=3) against the launch path (persistent=0): microseconds per iteration of one
trial (SG step-only, SG with E/Z/P, smoother mode) and trial-iterations/s of batches.
    python tools/ab_xcd.py [N ...]          (default 16 32 64)
(development aid; the committed logs are profiles/r03_xcd_*.log)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch  # noqa: E402
from solvers.spectral.batched import BatchedSGSolver  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [16, 32, 64]
K = int(os.environ.get("AB_K", "4096"))


def kw(N, mode, **extra):
    d = dict(name="spectral", Re=1000.0 if extra.pop("smoother", False) else 400.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5,
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
        for mode in (0, 3):
            s = SGSolver(**kw(N, mode, smoother=smoother))
            if smoother:
                s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
            s.run_iterations(256, diagnostics=diag)
            dt = timed(lambda: s.run_iterations(K, diagnostics=diag))
            row.append(dt / K * 1e6)
            s.close()
        print(f"N={N:3d} {label:14s}: launch path {row[0]:7.2f} us/iter   small-N kernel {row[1]:7.2f} us/iter   x{row[0] / row[1]:.2f}", flush=True)
    for B in (8, 16):
        row = []
        for mode in (0, 3):
            trials = [kw(N, mode, smoother=True, corner_smoothing=0.02 + 0.01 * q) for q in range(B)]
            b = BatchedSGSolver(trials)
            for s in b.solvers:
                s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
            b.run_iterations(128, diagnostics=False)
            dt = timed(lambda: b.run_iterations(K, diagnostics=False), reps=2)
            row.append(B * K / dt)
            b.close()
        print(f"N={N:3d} batch of {B:2d} smoothers: launch path {row[0] / 1e3:8.1f} k trial-it/s   small-N kernel {row[1] / 1e3:8.1f} k   x{row[1] / row[0]:.2f}", flush=True)
