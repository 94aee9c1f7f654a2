"""Iteration time (hipGraph path, N=256, with diagnostics) under the timing switches of ldc_debug_ablate: what a part
of the stage kernels costs THE ITERATION, not its own launch (development aid; the results of such runs are wrong on
purpose).    python tools/ab_masks.py 0 16 1 2 3 ..."""
import os
import sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
# timing switches live in the instrumented build only (-DLDC_TIMING, built by __graft_entry__.build())
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver

N = int(os.environ.get("AB_N", "256"))
diag = bool(int(os.environ.get("AB_DIAG", "1")))
for mask in [int(a) for a in sys.argv[1:]] or [0, 16]:
    s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
                 max_iterations=10**9, check_every=4096, graph_iters=64)
    s._ensure_handle(0.0)
    L.lib().ldc_debug_ablate(s._handle, mask)
    s.run_iterations(640, diagnostics=diag)
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); s.run_iterations(3200, diagnostics=diag); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 3200)
    print(f"mask {mask:6d}  N={N} diag={int(diag)}  {best:7.2f} us/iteration", flush=True)
    s.close()
