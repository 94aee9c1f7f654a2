"""Several single-trial solves on separate HIP streams against the same trials one after the other and against one
batched solve (development aid): trial-iterations per second at N (default 256).
    python tools/ab_streams.py [N] [n_trials]"""
import os
import sys
import time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver
from solvers.spectral.batched import BatchedSGSolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
K = 2048
kw = dict(name="spectral", nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0, max_iterations=10**9,
          check_every=4096, graph_iters=64)
sol = [SGSolver(Re=400.0 + 300.0 * q, **kw) for q in range(B)]
streams = [torch.cuda.Stream() for _ in range(B)]
for s, st in zip(sol, streams):
    with torch.cuda.stream(st):
        s.run_iterations(256, diagnostics=True)
torch.cuda.synchronize()


def timed(fn):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def one_after_the_other():
    for s in sol:
        L.check(L.lib().ldc_solver_enqueue(s._handle, K, 1, L.stream_ptr()))


def on_streams():
    for s, st in zip(sol, streams):
        with torch.cuda.stream(st):
            L.check(L.lib().ldc_solver_enqueue(s._handle, K, 1, L.stream_ptr()))


t_seq, t_par = timed(one_after_the_other), timed(on_streams)
print(f"N={N} B={B}: one after the other {B * K / t_seq:9.0f} trial-it/s   on {B} streams {B * K / t_par:9.0f} trial-it/s "
      f"({t_seq / t_par:.2f}x)", flush=True)
for s in sol:
    s.close()
b = BatchedSGSolver([dict(kw, Re=400.0 + 300.0 * q) for q in range(B)])
b.run_iterations(256, diagnostics=True)
t_b = timed(lambda: b.run_iterations(K, diagnostics=True))
print(f"N={N} B={B}: batched launches    {B * K / t_b:9.0f} trial-it/s", flush=True)
b.close()

# ---- batches on streams: S batched solves of b trials each, every batch on its own stream ------------------
if len(sys.argv) > 4:
    S, bsz = int(sys.argv[3]), int(sys.argv[4])
    bs = [BatchedSGSolver([dict(kw, Re=400.0 + 100.0 * (q + bsz * k)) for q in range(bsz)]) for k in range(S)]
    # AB_PRIO=1: the streams alternate between the two priorities HIP has -- streams of different priority never share
    # a hardware queue, streams of one priority may (which ones do is the runtime's choice)
    prio = bool(int(os.environ.get("AB_PRIO", "0")))
    junk = [torch.cuda.Stream() for _ in range(int(os.environ.get("AB_JUNK", "0")))]    # shifts the pool position
    junk += [torch.cuda.Stream(priority=-1) for _ in range(int(os.environ.get("AB_JUNK_HI", "0")))]
    sts = [torch.cuda.Stream(priority=(-1 if (prio and k % 2) else 0)) for k in range(S)]
    for x, st in zip(bs, sts):
        with torch.cuda.stream(st):
            x.run_iterations(256, diagnostics=True)
    torch.cuda.synchronize()

    def batches(parallel):
        for x, st in zip(bs, sts):
            if parallel:
                with torch.cuda.stream(st):
                    L.check(L.lib().ldc_batch_enqueue(x._batch, K, 1, L.stream_ptr()))
            else:
                L.check(L.lib().ldc_batch_enqueue(x._batch, K, 1, L.stream_ptr()))

    t0, t1 = timed(lambda: batches(False)), timed(lambda: batches(True))
    print(f"N={N}: {S} batches of {bsz}: one after the other {S * bsz * K / t0:9.0f}   on {S} streams {S * bsz * K / t1:9.0f} "
          f"trial-it/s ({t0 / t1:.2f}x)", flush=True)
    for x in bs:
        x.close()
