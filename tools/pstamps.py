"""Phase timing inside the persistent trial kernel (development aid): cycle stamps of the last iteration of a launch.
    python tools/pstamps.py N [diag] [mode: 1 anywhere, 2 one XCD]"""
import os
import sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np
# timing switches live in the instrumented build only (-DLDC_TIMING, built by __graft_entry__.build())
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
diag = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=4096, graph_iters=32, persistent=mode)
s.run_iterations(300, diagnostics=bool(diag))
nt = s.T * s.T
buf = torch.zeros(5 * nt * 64, dtype=torch.float64, device="cuda")
L.check(L.lib().ldc_debug_stamps(s._handle, buf.data_ptr()))
names = ["stage1", "sync1", "stage2", "sync2", "stage3", "sync3", "stage4", "sync4", "fin", "post"]
acc, inner = [], []
for rep in range(5):
    L.check(L.lib().ldc_solver_enqueue(s._handle, 200, diag, L.stream_ptr()))
    torch.cuda.synchronize()
    allst = buf.cpu().numpy().reshape(5, nt, 64)
    acc.append(np.diff(allst[0][:, :11], axis=1))
    inner.append(np.diff(allst[1:].reshape(4, nt, 8, 8)[..., :7], axis=-1))     # [stage, wg, wave, point]
d = np.median(np.stack(acc), axis=0)          # cycles, per work-group
di = np.median(np.stack(inner), axis=0)
print(f"N={N} T={s.T} diag={diag} mode={mode}: cycles per phase (median over 5 launches); mean / max over work-groups; us at 2.4 GHz")
for k, n in enumerate(names):
    print(f"  {n:8s} mean {d[:, k].mean():9.0f}  max {d[:, k].max():9.0f}   {d[:, k].mean() / 2400:7.2f} us")
print(f"  total    {d.sum(axis=1).mean():9.0f}   {d.sum(axis=1).mean() / 2400:7.2f} us (without the closing barrier)")
pts = ["entry->frags issued", "drain + K loop", "partials->LDS+barrier", "epilogue", "barrier", "stores+reductions"]
for k in range(4):
    print(f"  inside stage {k + 1} (mean over work-groups and waves / max), cycles:")
    for q, n in enumerate(pts):
        print(f"      {n:24s} {di[k, :, :, q].mean():9.0f} {di[k, :, :, q].max():9.0f}")
