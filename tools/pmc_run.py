#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes at N=256 (tools/pmc_run.py [N] [mode]).
mode -1 (default): the solver's own path -- the chip-wide kernel, ONE launch of 64 iterations (after the single launch-path
iteration that follows an upload); mode 0: 64 eager iterations of the launch path (the form of rounds 1-3)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
    sys.path.insert(0, p)
import __graft_entry__ as g
g.build()
from solvers.spectral.sg import SGSolver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = int(sys.argv[2]) if len(sys.argv) > 2 else -1
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=4096, graph_iters=4096, persistent=mode)   # graph never used: eager launches
s.run_iterations(1)          # the iteration after the upload (launch path; leaves phi^n with its boundary values)
s.run_iterations(64)         # mode 5: one launch of wide_kernel with 64 iterations
s.close()
print("done")
