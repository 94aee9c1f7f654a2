#!/usr/bin/env python3
"""Where the time goes INSIDE one stage launch: per-wave cycle stamps (ldc_debug_stamps).

Development aid.  Prints, per stage variant, the stamp times (us after the earliest wave start of the launch)
for the slowest tile, the median tile and the tiles that carry the index-(M-1) jobs.
points: 0 entry | 1 before the K loop | 2 after the K loop | 3 partials in LDS (barrier passed) |
        4 epilogue done | 5 barrier | 6 end
"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
# timing switches live in the instrumented build only (-DLDC_TIMING, built by __graft_entry__.build())
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

g.build()
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mask = int(os.environ.get("KB_MASK", "0"))
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=4096, graph_iters=32)
s.run_iterations(100)
lib, h, st = L.lib(), s._handle, L.stream_ptr()
T = s.T
buf = torch.zeros(T * T * 64, dtype=torch.float64, device="cuda")
lib.ldc_debug_stamps(h, buf.data_ptr())
lib.ldc_debug_ablate(h, 64 | mask)
MHZ = 2380.0


def tile(b):
    if T % 8 == 0:
        xcd, loc = b & 7, b >> 3
        pr, pc = T // 4, T // 2
        return (xcd >> 1) * pr + loc // pc, (xcd & 1) * pc + loc % pc
    return b // T, b % T


def job(b):
    i, j = tile(b)
    return ("R" if i == j else "") + ("C" if j == (i + 1) % T else "") + ("K" if (i, j) == ((0, 2) if T >= 3 else (0, 0)) else "")


for name, k in (("stage1", 1), ("stage0 GP", 0), ("stage3 LAST", 3), ("stage0 GP+omega", 16), ("stage1+grad omega", 17)):
    for _ in range(20):                      # warm: same kernel back to back like the bursts of kbench
        lib.ldc_stage(h, k, st)
    torch.cuda.synchronize()
    buf.zero_()
    lib.ldc_stage(h, k, st)
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(T * T, 8, 8)[:, :, :7]
    # the s_memtime counters of different blocks are not synchronised: only durations inside a block are used
    dur = (t - t[:, :, :1].min(axis=1, keepdims=True)) / MHZ
    total = dur[:, :, 6].max(axis=1)
    ej = np.array([bool(job(b)) for b in range(T * T)])
    print(f"== {name}: block duration median {np.median(total):.2f} p90 {np.quantile(total, .9):.2f} max {total.max():.2f}")
    for label, sel in (("tiles without an M-1 job", ~ej), ("tiles with an M-1 job", ej)):
        if not sel.any():
            continue
        print(f"   {label} ({int(sel.sum())}): duration median {np.median(total[sel]):.2f} max {total[sel].max():.2f}; "
              "median per-wave stamps (wave 0..7 = role0 kq0..3, role1 kq0..3):")
        for w in range(8):
            print(f"     wave {w}: " + " ".join(f"{x:6.2f}" for x in np.median(dur[sel][:, w], axis=0)))
    worst = np.argsort(total)[-6:][::-1]
    print("   longest blocks:", ", ".join(f"{b}{tuple(int(v) for v in tile(b))}{job(b)}:{total[b]:.2f}" for b in worst))
lib.ldc_debug_ablate(h, 0)
lib.ldc_debug_stamps(h, None)
s.close()
