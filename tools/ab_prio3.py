"""Three worker streams on the three stream priorities HIP has (torch only offers two): trial-iterations per second of
three single N=256 trials / three batches at N=128, against two streams (development aid).
    python tools/ab_prio3.py N n_per_batch [n_batches=3]"""
import ctypes
import os
import sys
import time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver
from solvers.spectral.batched import BatchedSGSolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.cuda.init(); torch.zeros(1, device="cuda")
path = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0]
hip = ctypes.CDLL(path)                      # the copy torch has loaded, not a second runtime
lo, hi = ctypes.c_int(), ctypes.c_int()
assert hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0
print("stream priority range: least", lo.value, "greatest", hi.value, flush=True)


def stream(prio):
    h = ctypes.c_void_p()
    assert hip.hipStreamCreateWithPriority(ctypes.byref(h), 1, prio) == 0      # 1 = hipStreamNonBlocking
    return torch.cuda.ExternalStream(h.value)


K = 2048
kw = dict(name="spectral", nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0, max_iterations=10**9,
          check_every=4096, graph_iters=64)


def make(k):
    if b == 1:
        s = SGSolver(Re=400.0 + 100.0 * k, **kw)
        s.run_iterations(256, diagnostics=True)
        return s, (lambda: L.check(L.lib().ldc_solver_enqueue(s._handle, K, 1, L.stream_ptr())))
    x = BatchedSGSolver([dict(kw, Re=400.0 + 100.0 * (q + b * k)) for q in range(b)])
    x.run_iterations(256, diagnostics=True)
    return x, (lambda: L.check(L.lib().ldc_batch_enqueue(x._batch, K, 1, L.stream_ptr())))


objs = [make(k) for k in range(nb)]
torch.cuda.synchronize()


def timed(streams, which):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in which:
            with torch.cuda.stream(streams[k % len(streams)]):
                objs[k][1]()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


two = [stream(0), stream(hi.value)]
three = [stream(lo.value), stream(0), stream(hi.value)]
t1 = timed([torch.cuda.current_stream()], list(range(nb)))
t2 = timed(two, list(range(nb)))
t3 = timed(three, list(range(nb)))
print(f"N={N} {nb} x {b} trial(s): one stream {nb * b * K / t1:9.0f}   two priorities {nb * b * K / t2:9.0f}   "
      f"three priorities {nb * b * K / t3:9.0f} trial-it/s", flush=True)
