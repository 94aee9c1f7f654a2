"""FSG solve (smoother mode: a transform launch after every stage, eight launches per iteration) on the launch path
and with the persistent trial kernel, anywhere (1) and on one XCD (2) -- development aid.
    python tools/ab_fsg_persist.py"""
import os, sys, time
# modes 1 and 2 live in the instrumented build only (csrc/ldc_trial_kernel.inc, -DLDC_TIMING)
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral.fsg import FSGSolver
for mode in (0, 2, 1):
    s = FSGSolver(name="spectral_fsg", Re=1000.0, nx=64, ny=64, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                  corner_treatment="smoothing", corner_smoothing=0.15, multigrid="fsg", n_levels=2,
                  coarse_tolerance_factor=10.0, tolerance=1e-9, max_iterations=20000, check_every=2048, graph_iters=64,
                  persistent=mode)
    torch.cuda.synchronize(); t0 = time.perf_counter(); s.solve(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"persistent={mode}: {s.metrics.iterations} iterations in {dt:.2f} s = {dt / s.metrics.iterations * 1e6:.1f} us/iteration", flush=True)
    s.close()
