"""A/B timing of the full iteration (hipGraph path, N=256): run once per library, e.g.
    python tools/ab_iter.py; LDC_HIP_LIB=/path/to/other/libldc_hip.so python tools/ab_iter.py
(development aid)"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
if "AB_ABLATE" in os.environ:      # the switches live in the instrumented build only (-DLDC_TIMING)
    os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver
N = int(os.environ.get("AB_N", "256"))
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=4096, graph_iters=32)
if os.environ.get("AB_ABLATE"):      # e.g. 128 plain stores, 256 write-through stores (before the graph is built)
    s._ensure_handle(0.0)
    L.lib().ldc_debug_ablate(s._handle, int(os.environ["AB_ABLATE"]))
s.run_iterations(640)
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); s.run_iterations(3200); e1.record(); torch.cuda.synchronize()
    print(os.environ.get("LDC_HIP_LIB", "main")[-24:], f"N={N} ablate={os.environ.get('AB_ABLATE', '0')}",
          f"{e0.elapsed_time(e1) * 1e3 / 3200:.2f} us/iter")
