"""Feasibility probe: the step()-only loop on one stream while a second (low-priority) stream runs one stand-alone
omega / palinstrophy pass per iteration -- would per-iteration diagnostics hide in the launch gaps of the main chain?
(development aid; the side passes here also recompute the pressure transforms, which is harmless for timing only)
    python tools/ab_side_diag.py"""
import ctypes as C
import os
import sys
import threading
import time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver

N, K = 256, 4096
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=8192, graph_iters=64)
s.run_iterations(512, diagnostics=True)
s.run_iterations(512, diagnostics=False)
lib, h = L.lib(), s._handle


def stream(prio):
    hnd = C.c_void_p()
    L.check(lib.ldc_stream_create(prio, C.byref(hnd)))
    return torch.cuda.ExternalStream(hnd.value)


def loop(diag):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    L.check(lib.ldc_solver_enqueue(h, K, diag, L.stream_ptr()))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6


print(f"fused diagnostics      {min(loop(1) for _ in range(3)):7.2f} us/iteration", flush=True)
print(f"step only              {min(loop(0) for _ in range(3)):7.2f} us/iteration", flush=True)
for prio, name in ((1, "low"), (0, "normal")):
    side = stream(prio)
    done = {"n": 0}
    stop = threading.Event()

    def feeder():
        with torch.cuda.stream(side):
            while not stop.is_set():
                for _ in range(64):
                    lib.ldc_diagnostics(h, L.stream_ptr())
                done["n"] += 64
                side.synchronize()

    best = 1e9
    for _ in range(3):
        done["n"] = 0; stop.clear()
        th = threading.Thread(target=feeder); th.start()
        time.sleep(0.02)
        n0 = done["n"]
        t = loop(0)
        n1 = done["n"]
        stop.set(); th.join()
        best = min(best, t)
        rate = (n1 - n0) / K
    print(f"step only + side passes on a {name}-priority stream: {best:7.2f} us/iteration, {rate:.2f} side passes per iteration",
          flush=True)
s.close()
