"""A/B timing of a HETEROGENEOUS batch: B trials, all but one with a loose tolerance (they latch early), the last
one runs on.  A batch should get cheaper as its trials converge.  AB_N, AB_B, LDC_HIP_LIB.  (development aid)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral.batched import BatchedSGSolver
N, B = int(os.environ.get("AB_N", "128")), int(os.environ.get("AB_B", "4"))
trials = [dict(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, corner_smoothing=0.15,
               tolerance=(1e-2 if q < B - 1 else 0.0), max_iterations=10**9, check_every=512, graph_iters=32)
          for q in range(B)]
b = BatchedSGSolver(trials)
out = b.run_to_tolerance([t["tolerance"] for t in trials], 2048)            # warm-up, the loose ones latch here
print("latched after warm-up:", [int(d) for d, _, _ in out], "iterations", [int(t) for _, t, _ in out])
torch.cuda.synchronize(); t0 = time.perf_counter()
b._advance(512, True); b._advance(512, True); b._advance(512, True); b._advance(512, True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(os.environ.get("LDC_HIP_LIB", "main")[-20:], f"N={N} B={B} (1 live trial): {dt / 2048 * 1e6:.2f} us/iteration")
b.close()
