"""Where the time goes inside ONE iteration of the trial-per-CU kernel (persistent=4): per-wave cycle stamps of the last
iteration of a chunk (instrumented build, ldc_debug_stamps), per RK stage.
    python tools/cstamps.py [N] [sg|diag|smoother] [B]
Points: 0 stage entry | 6 extra duty done (fold wave / ring wave) | 1 contractions (tile waves) / dot products (edge waves)
done | 2 barrier passed | 3 epilogue done | 4 sums reduced | 5 barrier passed = next stage entry.
B > 1: that many trials in one batch launch (work-group 0 is printed): the chip under load.
(development aid; logs: profiles/r03_cu_stamps_*.log)"""
import os
import sys

os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.batched import BatchedSGSolver  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
kind = sys.argv[2] if len(sys.argv) > 2 else "sg"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
trials = [dict(name="spectral", Re=1000.0 if kind == "smoother" else 400.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
               max_iterations=10**9, check_every=1024, graph_iters=64, persistent=4, corner_smoothing=0.05 + 0.0005 * q) for q in range(B)]
b = BatchedSGSolver(trials)
if kind == "smoother":
    for s in b.solvers:
        s._stage_pressure, s._warmup, s._nan_exit = 1, 0, True
diag = kind == "diag"
b.run_iterations(256, diagnostics=diag)
if any(int(L.lib().ldc_solver_mode(s._handle)) != 4 for s in b.solvers):
    raise SystemExit(f"N={N}: the trial-per-CU kernel does not apply in this build (3 x 3 tiles, M = 34 ... 44: the LDS layout of that class leaves no room for the stamps)")
W, P = 11, 8
buf = torch.zeros(B * W * 4 * P, dtype=torch.float64, device="cuda")
for q, s in enumerate(b.solvers):
    L.check(L.lib().ldc_debug_stamps(s._handle, buf.data_ptr()), "ldc_debug_stamps")
b.close_batch()                 # argument blocks are made at batch creation: make them again with the stamp pointer
b.run_iterations(512, diagnostics=diag)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(B, W, 4, P)[0]
M = N + 1
edge = M >= 17 and (M - 1) % 16 == 0
T = (M - 1) // 16 if edge else (M + 15) // 16
nw = T * T + (4 if (edge or T < 3) else 0)      # four helper waves (EDGE mode: two of them are the edge waves)
print(f"N={N} ({T}x{T} tile waves{' + 2 edge + 2 helper waves' if edge else (' + 4 helper waves' if T < 3 else '')}), {kind}, batch of {B}: cycles (s_memtime) relative to the stage entry of wave 0")
names = {0: "entry", 6: "duty", 1: "products", 2: "barrier", 3: "epilogue", 4: "sums", 5: "barrier"}
for k in range(4):
    base = st[0, k, 0]
    print(f"-- stage {k + 1}: length (wave 0 entry -> exit) {st[0, k, 5] - base:.0f} cycles")
    for wv in range(nw):
        print(f"   wave {wv:2d}: " + "  ".join(f"{names[p]}={st[wv, k, p] - base:6.0f}" for p in (0, 6, 1, 2, 3, 4, 5)))
print(f"iteration (stage 1 entry -> stage 4 exit): {st[0, 3, 5] - st[0, 0, 0]:.0f} cycles")
