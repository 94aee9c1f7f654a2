"""Trial-per-CU kernel (persistent=4) against the small-N kernel (persistent=3) and the launch path on BATCHES of equal-N
trials: trial-iterations per second by N and batch size (step()-only loop; `diag` as third argument: with E/Z/P).
    python tools/ab_cu.py [N,N,...] [B,B,...] [sg|diag|smoother]
(development aid; log: profiles/r03_cu_ab.log)"""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.batched import BatchedSGSolver  # noqa: E402

Ns = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,30,32,40").split(",")]
Bs = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "8,64,256").split(",")]
kind = sys.argv[3] if len(sys.argv) > 3 else "sg"
diag = kind == "diag"


def rate(N, B, mode, K):
    # (Re = 100 ...: at N <= 40 the scheme itself diverges at Re = 1000 within a few thousand iterations -- the oracle too)
    trials = [dict(name="spectral", Re=100.0 + 0.5 * q, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=0.0,
                   max_iterations=10**9, basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing",
                   corner_smoothing=0.02 + 0.0005 * q, multigrid="none", check_every=K, graph_iters=64, persistent=mode)
              for q in range(B)]
    b = BatchedSGSolver(trials)
    if kind == "smoother":
        for s in b.solvers:
            s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
    b.run_iterations(64, diagnostics=diag)
    best = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.run_iterations(K, diagnostics=diag)
        torch.cuda.synchronize()
        best = max(best, B * K / (time.perf_counter() - t0))
    modes = {int(L.lib().ldc_solver_mode(s._handle)) for s in b.solvers}
    if kind != "smoother":
        assert all(bool(torch.isfinite(s.d["rec"]).all()) for s in b.solvers), "a trial diverged"
    b.close()
    return best, modes


print(f"trial-iterations/s, {kind}; columns: launch path | small-N kernel (one trial per XCD) | trial per CU")
for N in Ns:
    for B in Bs:
        K = 4096         # (a chunk of a sweep: check_every iterations per launch)
        row = []
        for mode in (0, 3, 4):
            if mode == 0 and B > 64:
                row.append("      -     ")
                continue
            r, modes = rate(N, B, mode, K)
            row.append(f"{r / 1e6:8.3f} M ({','.join(str(m) for m in sorted(modes))})")
        print(f"N={N:3d} B={B:4d}: " + " | ".join(row), flush=True)
