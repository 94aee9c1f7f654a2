"""Soak of the chip-wide kernel: two solvers with the same inputs side by side, long; every chunk's history rows, state arrays and
partial-sum slabs must agree bit for bit (a protocol hole shows as a timing-dependent difference).
    python tools/debug/soak_wide.py [N ...]      LDC_WIDE_LAYOUT=tail|tiles as usual"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np
from solvers.spectral.sg import SGSolver
from solvers.spectral import ldc_lib as L
sizes = [int(x) for x in sys.argv[1:]] or [256]
CH, NCH = int(os.environ.get("SOAK_CHUNK", "509")), int(os.environ.get("SOAK_CHUNKS", "160"))
for N in sizes:
    kw = dict(name="spectral", Re=1000.0, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=1e-30,
              max_iterations=10**9, basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing",
              corner_smoothing=0.15, multigrid="none", check_every=1024, graph_iters=16, persistent=5)
    A, B = SGSolver(**kw), SGSolver(**kw)
    bad, t0 = 0, time.perf_counter()
    for c in range(NCH):
        ra, rb = A.run_iterations(CH), B.run_iterations(CH)
        ok = np.array_equal(ra, rb)
        for key in ("partials", "U", "V", "P"):
            ok = ok and np.array_equal(A.d[key].cpu().numpy(), B.d[key].cpu().numpy())
        if not ok:
            bad += 1
            print(f"N={N} chunk {c}: DIFFERENT", flush=True)
    assert L.lib().ldc_solver_mode(A._handle) == 5
    print(f"N={N} layout={os.environ.get('LDC_WIDE_LAYOUT', 'default')}: {NCH} chunks of {CH} iterations twice, {bad} differing, "
          f"{time.perf_counter() - t0:.1f} s, finite={bool(np.all(np.isfinite(ra)))}", flush=True)
    A.close(); B.close()
