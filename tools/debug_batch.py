import sys, time, faulthandler
from pathlib import Path
faulthandler.dump_traceback_later(70, exit=True)
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
    sys.path.insert(0, p)
def say(*a):
    print(*a, flush=True)
import torch
import __graft_entry__ as g
g.build()
from solvers.spectral.batched import BatchedSGSolver
from solvers.spectral import ldc_lib as L
N = 16
tols = [1e-3, 1e-4, 3e-4]
trials = [dict(name="spectral", Re=100.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=t,
               max_iterations=10**9, check_every=256, graph_iters=8) for t in tols]
b = BatchedSGSolver(trials)
b._ensure_batch(tols)
for s in b.solvers:
    s.d["ctrl"].zero_(); s._prime()
torch.cuda.synchronize(); say("primed")
lib = L.lib()
diag = int(sys.argv[1]) if len(sys.argv) > 1 else 1
def ctrls():
    return [s.d["ctrl"].cpu().tolist()[:4] for s in b.solvers]
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prev = None
for it in range(0, 4000, chunk):
    lib.ldc_batch_enqueue(b._batch, chunk, diag, L.stream_ptr())
    torch.cuda.synchronize()
    c = ctrls()
    key = tuple(x[0] for x in c)
    if key != prev or it % 512 == 0:
        say("after", it + chunk, c)
        prev = key
    if all(key):
        break
say("done")
