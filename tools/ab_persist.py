"""Persistent trial kernel (mode 2: all work-groups on one XCD; mode 1: anywhere) vs launch-per-stage path: microseconds per iteration by N (development aid).
    python tools/ab_persist.py [N ...]        env: AB_DIAG=0/1 (default 1), AB_ITERS"""
import os
import sys
# modes 1 and 2 live in the instrumented build only (csrc/ldc_trial_kernel.inc, -DLDC_TIMING)
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral import ldc_lib as L
from solvers.spectral.sg import SGSolver

sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128]
diag = bool(int(os.environ.get("AB_DIAG", "1")))
iters = int(os.environ.get("AB_ITERS", "2048"))
for N in sizes:
    row = []
    for mode in (2, 1, 0):
        try:
            s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
                         max_iterations=10**9, check_every=4096, graph_iters=32, persistent=mode)
            s.run_iterations(256, diagnostics=diag)
            best = 1e30
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                L.check(L.lib().ldc_solver_enqueue(s._handle, iters, int(diag), L.stream_ptr()))
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
            assert int(s.d["sync"][L.SYNC_GIVEUP]) == 0, "a barrier wait was given up"
            row.append(best)
            s.close()
        except Exception as exc:          # e.g. persistent not available at this size
            row.append(float("nan"))
            print(f"N={N} mode={mode}: {exc}", flush=True)
    print(f"N={N:4d} diag={int(diag)}  one-XCD {row[0]:8.2f}   persistent {row[1]:8.2f}   launches {row[2]:8.2f} us/iter   "
          f"launches / one-XCD {row[2] / row[0]:.2f}x   launches / persistent {row[2] / row[1]:.2f}x", flush=True)
