"""A/B timing of the batched path: AB_N, AB_B, AB_ABLATE (128 plain / 256 write-through stores), LDC_HIP_LIB.
(development aid)"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
if "AB_ABLATE" in os.environ:      # the switches live in the instrumented build only (-DLDC_TIMING)
    os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.batched import BatchedSGSolver
N, B = int(os.environ.get("AB_N", "32")), int(os.environ.get("AB_B", "1"))
trials = [dict(name="spectral", Re=100.0 + 50 * q, nx=N, ny=N, basis_type="chebyshev", CFL=1.5,
               corner_smoothing=0.05 + 0.01 * (q % 20), tolerance=0.0, max_iterations=10**9,
               check_every=1024, graph_iters=32) for q in range(B)]
b = BatchedSGSolver(trials)
if os.environ.get("AB_ABLATE"):
    for s in b.solvers:
        s._ensure_handle(0.0)
        L.lib().ldc_debug_ablate(s._handle, int(os.environ["AB_ABLATE"]))
b.run_iterations(256)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b.run_iterations(1024)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(os.environ.get("LDC_HIP_LIB", "main")[-20:], f"N={N} B={B} ablate={os.environ.get('AB_ABLATE', '0')}: {dt / 1024 * 1e6:.2f} us/iteration")
b.close()
