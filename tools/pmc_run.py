#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 --pmc passes: 64 eager iterations at N=256."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
    sys.path.insert(0, p)
import __graft_entry__ as g
g.build()
from solvers.spectral.sg import SGSolver
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=4096, graph_iters=4096)   # graph never used: eager launches
s.run_iterations(64)
s.close()
print("done")
