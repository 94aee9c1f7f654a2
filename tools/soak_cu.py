"""Soak of the trial-per-CU kernel: the SAME batch of trials advanced twice from rest, every record row and every state array
compared bit for bit (the helper waves' duties, the exchange through LDS tables and the fold run beside the contractions: a
race would show as a difference between two identical runs).
    python tools/soak_cu.py [N,N,...] [trials] [iterations]        (log: profiles/r04_cu_soak.log)"""
import os
import sys

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.batched import BatchedSGSolver  # noqa: E402

Ns = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,30,32,40").split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40960


def run(N, diag):
    trials = [dict(name="spectral", Re=100.0 + 0.5 * q, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=0.0,
                   max_iterations=10**9, basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing",
                   corner_smoothing=0.02 + 0.0005 * q, multigrid="none", check_every=4096, graph_iters=64, persistent=4)
              for q in range(B)]
    b = BatchedSGSolver(trials)
    recs = []
    for _ in range(K // 4096):
        b.run_iterations(4096, diagnostics=diag)
        recs.append(torch.stack([s.d["rec"].clone() for s in b.solvers]))
    torch.cuda.synchronize()
    assert all(int(L.lib().ldc_solver_mode(s._handle)) == 4 for s in b.solvers)
    state = torch.stack([torch.stack([s.d[k].clone() for k in ("U", "V", "P")]) for s in b.solvers])
    b.close()
    return torch.cat(recs, dim=1), state


for N in Ns:
    for diag in (False, True):
        r1, s1 = run(N, diag)
        r2, s2 = run(N, diag)
        same_r = bool(torch.equal(r1, r2)) or bool(((r1 == r2) | (torch.isnan(r1) & torch.isnan(r2))).all())
        same_s = bool(torch.equal(s1, s2))
        print(f"N={N:3d} {'E/Z/P' if diag else 'step only'}: {B} trials x {K} iterations twice: records equal {same_r}, states equal {same_s}; "
              f"finite {bool(torch.isfinite(r1).all())}", flush=True)
        assert same_r and same_s
