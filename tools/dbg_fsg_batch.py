"""One batched FSG solve of B equal-N trials (development aid: isolates which size / kernel a failure of a large search round
belongs to).    python tools/dbg_fsg_batch.py N B [Re] [max_iterations]"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.batched import BatchedFSGSolver  # noqa: E402

N, B = int(sys.argv[1]), int(sys.argv[2])
Re = float(sys.argv[3]) if len(sys.argv) > 3 else 1000.0
cap = int(sys.argv[4]) if len(sys.argv) > 4 else 20000
trials = [dict(name="spectral_fsg", Re=Re, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=1e-6, max_iterations=cap,
               basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.01 + 0.09 * q / B,
               multigrid="fsg", n_levels=2, coarse_tolerance_factor=1.0, prolongation_method="fft", restriction_method="fft",
               check_every=2048, graph_iters=64) for q in range(B)]
b = BatchedFSGSolver(trials)
ms = b.solve()
torch.cuda.synchronize()
print(f"N={N} B={B} Re={Re:g}: levels {b.orders}, iterations {sorted({m.iterations for m in ms})[:4]} ..., converged {sum(m.converged for m in ms)} of {B}", flush=True)
b.close()
