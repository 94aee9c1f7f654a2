"""Iterations captured per hipGraph against microseconds per iteration at N=256 (development aid).
    python tools/ab_graph_iters.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver
for gi in (32, 64, 128, 256, 512):
    s = SGSolver(name="spectral", Re=1000.0, nx=256, ny=256, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
                 max_iterations=10**9, check_every=8192, graph_iters=gi)
    s.run_iterations(1024)
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        L.check(L.lib().ldc_solver_enqueue(s._handle, 4096, 1, L.stream_ptr()))
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 4096)
    print(f"graph_iters={gi}: {best:.3f} us/iteration", flush=True)
    s.close()
