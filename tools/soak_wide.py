"""Soak of the chip-wide trial kernel (persistent=MODE): two solvers with the same inputs advanced side by side from rest in
chunks of an odd number of iterations; every history row, every state array and every slab of partial sums compared bit for
bit after each chunk (the kernel is a protocol between work-groups that meet only through flags: a hole in it shows as a
result that depends on timing -- DESIGN.md 3, "MFMA work in a window was not reproducible").
    python tools/soak_wide.py [N,N,...] [chunks] [iterations per chunk] [layout,layout,...] [mode]   (log: profiles/r04_wide_soak_long.log)
mode 3 instead of 5: the same soak of the one-XCD trial kernel (N <= 79; log: profiles/r04_xcd_soak.log)."""
import os
import sys
import time

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402

Ns = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,255,200,128,96,81").split(",")]
CHUNKS = int(sys.argv[2]) if len(sys.argv) > 2 else 400
PER = int(sys.argv[3]) if len(sys.argv) > 3 else 2039
LAYOUTS = (sys.argv[4] if len(sys.argv) > 4 else "default").split(",")
MODE = int(sys.argv[5]) if len(sys.argv) > 5 else 5


def make(N):
    s = SGSolver(name="spectral", Re=1000.0 if N >= 40 else 100.0,      # (Re = 1000 diverges on the coarsest grids, in the oracle too)
                 lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5,
                 beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15, tolerance=1e-6, max_iterations=10**9,
                 multigrid="none", check_every=4096, graph_iters=64, persistent=MODE)
    s._ensure_handle(0.0)
    return s


bad = 0
for layout in LAYOUTS:
    if layout != "default":
        os.environ["LDC_WIDE_LAYOUT"] = layout
    for N in Ns:
        A, B = make(N), make(N)
        if int(L.lib().ldc_solver_mode(A._handle)) != MODE or int(L.lib().ldc_solver_mode(B._handle)) != MODE:
            print(f"N={N} layout={layout}: not on kernel mode {MODE}, skipped", flush=True)
            A.close(); B.close()
            continue
        t0, differing, finite = time.perf_counter(), 0, True
        for c in range(CHUNKS):
            ra, rb = A.run_iterations(PER), B.run_iterations(PER)
            same = np.array_equal(ra, rb, equal_nan=True) and all(
                bool(((A.d[k] == B.d[k]) | (torch.isnan(A.d[k]) & torch.isnan(B.d[k]))).all()) for k in ("partials", "U", "V", "P") if k in A.d)
            differing += 0 if same else 1
            finite = finite and bool(np.isfinite(ra).all())
        print(f"N={N} mode={MODE} layout={layout}: {CHUNKS} chunks of {PER} iterations twice, {differing} differing, "
              f"{time.perf_counter() - t0:.1f} s, finite={finite}", flush=True)
        bad += differing
        A.close(); B.close()
sys.exit(1 if bad else 0)
