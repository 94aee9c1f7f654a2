"""Timing probe for the forked-graph idea: the N=256 iteration with the post launch on a side branch BESIDE
stage 4 (instrumented build, LDC_FORK_PROBE=1: results are wrong, the timing is what counts) against the serial graph.
    python tools/ab_fork_probe.py          (runs both arms in child processes, alternating)
(development aid; log: profiles/r03_fork_probe.log)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TIMING = os.path.join(ROOT, "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so")
CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch
from solvers.spectral.sg import SGSolver
s = SGSolver(name="spectral", Re=1000.0, nx=256, ny=256, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=4096, graph_iters=64, persistent=0)
for diag in (True, False):
    s.run_iterations(640, diagnostics=diag)
    best = 1e9
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); s.run_iterations(3200, diagnostics=diag); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 3200)
    print(("forked " if os.environ.get("LDC_FORK_PROBE") == "1" else "serial ") + ("with E/Z/P" if diag else "step only ") + f": {best:.2f} us/iteration", flush=True)
'''
for arm in ("0", "1", "0", "1"):
    env = dict(os.environ, LDC_HIP_LIB=TIMING, LDC_FORK_PROBE=arm)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    print(r.stdout.strip() or r.stderr[-800:], flush=True)
