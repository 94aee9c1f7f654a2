"""Where the time goes inside ONE iteration of the chip-wide trial kernel (persistent=5): per-wave cycle stamps of the last
iteration of a chunk (instrumented build, ldc_debug_stamps), printed per RK stage as medians over the work-groups.
    python tools/wstamps.py [N] [smoother|sg|diag]
Points: 0 stage entry | 1 contractions done (waves 0-3) | 2 this wave's part of the contraction phase done (results in LDS,
ring, fold) | 3 work-group barrier passed | 4 epilogue done | 5 barrier passed (stages with sums) | 6 sums / partials stored |
7 stores drained (vmcnt 0) | 8 barrier | 9 mates' flags seen (wave 0) | 10 barrier passed = next stage entry.
(development aid; logs: profiles/r04_wide_stamps_*.log)"""
import os
import sys

# timing switches live in the instrumented build only (-DLDC_TIMING, built by __graft_entry__.build())
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
kind = sys.argv[2] if len(sys.argv) > 2 else "diag"
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5,
             tolerance=0.0, max_iterations=10**9, check_every=1024, graph_iters=64, persistent=5)
if kind == "smoother":
    s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
diag = kind == "diag"
s.run_iterations(256, diagnostics=diag)
assert L.lib().ldc_solver_mode(s._handle) == 5
_tiles_fit = ((s.M + 15) // 16) ** 2 <= 256
_lay = os.environ.get("LDC_WIDE_LAYOUT", "tiles" if _tiles_fit else "tail")
T = (s.M - 1) // 16 if ((s.M - 1) % 16 == 0 and kind != "smoother" and (_lay == "tail" or not _tiles_fit)) else (s.M + 15) // 16
nwg, W, P = T * T, 8, 12
buf = torch.zeros(nwg * W * 4 * P, dtype=torch.float64, device="cuda")
L.check(L.lib().ldc_debug_stamps(s._handle, buf.data_ptr()), "ldc_debug_stamps")
s.run_iterations(512, diagnostics=diag)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(nwg, W, 4, P)
L.lib().ldc_debug_stamps(s._handle, None)
print(f"N={N} ({nwg} work-groups), {kind}: cycles (s_memtime) relative to the stage entry of wave 0 of each work-group; median [min..max] over work-groups")
names = ["entry", "contr|fold:flags", "phase done", "barrier", "epilogue", "barrier", "sums out", "drained", "barrier", "flags seen|fold:loads", "next entry", "fold:totals"]
tail = T * 16 + 1 == s.M
ii, jj = np.arange(nwg) // T, np.arange(nwg) % T
jobs = ((ii == jj) | (jj == (ii + 1) % T) | ((ii == 1) & (jj == 3))) if tail else np.zeros(nwg, bool)
groups = [("all", np.ones(nwg, bool))] + ([("job tiles", jobs), ("other tiles", ~jobs)] if tail else [])
for k in range(4):
    base = st[:, 0, k, 0][:, None]
    print(f"-- stage {k + 1}: length (wave 0 entry -> next entry) median {np.median(st[:, 0, k, 10] - st[:, 0, k, 0]):.0f} cycles")
    for gname, sel in groups:
        if len(groups) > 1:
            print(f"  [{gname}: {int(sel.sum())}]")
        for wv in range(8):
            row = []
            for p in range(12):
                ok = sel & (st[:, wv, k, p] > 0)
                x = (st[:, wv, k, p] - base[:, 0])[ok]
                row.append("-" if x.size == 0 else f"{np.median(x):.0f}[{x.min():.0f}..{x.max():.0f}]")
            print(f"   wave {wv}: " + "  ".join(f"{n}={r}" for n, r in zip(names, row)))
tot = st[:, 0, 3, 10] - st[:, 0, 0, 0]
print(f"iteration (stage 1 entry -> stage 4 exit): median {np.median(tot):.0f} cycles")
