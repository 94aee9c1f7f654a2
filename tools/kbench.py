#!/usr/bin/env python3
"""Per-kernel timing of the hot path with HIP events (bursts of identical launches).

Development aid (not part of the product path).  Usage:
    python tools/kbench.py [--N 256] [--reps 200]
"""
import argparse
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
    sys.path.insert(0, p)

# timing switches live in the instrumented build only (-DLDC_TIMING, built by __graft_entry__.build())
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

g.build()
from solvers.spectral import ldc_lib as L  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=256)
ap.add_argument("--Re", type=float, default=1000.0)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--batched", action="store_true")
a = ap.parse_args()
s = SGSolver(name="spectral", Re=a.Re, nx=a.N, ny=a.N, basis_type="chebyshev", CFL=1.5, tolerance=0.0,
             max_iterations=10**9, check_every=4096, graph_iters=32)
s.run_iterations(300)          # a developed (non-trivial) state
lib, h, st = L.lib(), s._handle, L.stream_ptr()


def burst(fn, reps=a.reps, rounds=7):
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


M = a.N + 1
rows = [("stage0 (GP)", lambda: lib.ldc_stage(h, 0, st), 20 * M**3),
        ("stage1", lambda: lib.ldc_stage(h, 1, st), 16 * M**3),
        ("stage2", lambda: lib.ldc_stage(h, 2, st), 16 * M**3),
        ("stage3 (LAST)", lambda: lib.ldc_stage(h, 3, st), 16 * M**3),
        ("stage0 (GP) + omega", lambda: lib.ldc_stage(h, 16, st), 20 * M**3),
        ("stage1 + grad omega", lambda: lib.ldc_stage(h, 17, st), 20 * M**3),
        ("ptrans", lambda: lib.ldc_pressure_transform(h, 0, st), 4 * M**3),
        ("diagnostics (post+omega, palin)", lambda: lib.ldc_diagnostics(h, st), 12 * M**3),
        ("finalize", lambda: lib.ldc_finalize(h, 1, st), 0)]
# tolerance is 0, so stage 3 / finalize never latch
for name, fn, fl in rows:
    med, best = burst(fn)
    tf = fl / med / 1e6 if fl else 0.0
    print(f"{name:34s} median {med:8.2f} us  best {best:8.2f} us  {tf:6.2f} TFLOP/s")
extra = [(int(m), f"mask {m}") for m in os.environ.get("KB_MASKS", "").split(",") if m]
for mask, label in [(1, "no MFMA"), (2, "no operand loads"), (3, "neither")] + extra:
    lib.ldc_debug_ablate(h, mask)
    for name, fn, fl in rows[:6]:
        med, best = burst(fn)
        print(f"  ablate[{label:16s}] {name:16s} median {med:8.2f} us")
lib.ldc_debug_ablate(h, 0)
s.reset_state(); s.run_iterations(50)
# fp64 MFMA issue rate and the clock it runs at (s_memtime vs the 100 MHz s_memrealtime)
grid, iters = 256 * 4, 20000
sink = torch.zeros(grid * 256 + 8, dtype=torch.float64, device="cuda")
for g_, label in ((256, "1 wave/SIMD"), (512, "2 waves/SIMD"), (1024, "4 waves/SIMD, VGPR-limited to 3")):
    lib.ldc_mfma_peak(sink.data_ptr(), 200, g_, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.ldc_mfma_peak(sink.data_ptr(), iters, g_, st); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    cyc, ticks = float(sink[0]), float(sink[1])
    tf = g_ * 4 * iters * 8 * 2048.0 / (ms * 1e-3) / 1e12
    print(f"mfma_peak grid={g_:5d} ({label}): {tf:6.2f} TFLOP/s, {ms*1e3:8.1f} us, block0: {cyc/(iters*8):6.1f} cyc/MFMA/wave, "
          f"clock {cyc/ticks*100:7.1f} MHz")
for nd in (0, 1):
    med, best = burst(lambda: lib.ldc_solver_enqueue(h, 32, nd, st), reps=10)
    print(f"graph iteration diag={nd}: {med / 32:8.2f} us/iter  ({32e6 / med:9.1f} it/s)")
s.close()

# ---- batched trials: aggregate trial-iterations/s when B trials share every launch -------------
if "--batched" in sys.argv:
    from solvers.spectral.batched import BatchedSGSolver
    for N, Bs in ((32, (1, 16, 64)), (64, (1, 4, 16)), (128, (1, 2, 4))):
        for B in Bs:
            trials = [dict(name="spectral", Re=100.0 + 50 * q, nx=N, ny=N, basis_type="chebyshev", CFL=1.5,
                           corner_smoothing=0.05 + 0.01 * (q % 20), tolerance=0.0, max_iterations=10**9,
                           check_every=1024, graph_iters=32) for q in range(B)]
            b = BatchedSGSolver(trials)
            b.run_iterations(200)
            t0 = __import__("time").perf_counter()
            b.run_iterations(1024)
            dt = __import__("time").perf_counter() - t0
            print(f"batched N={N:4d} B={B:3d}: {dt / 1024 * 1e6:8.2f} us/iteration, {B * 1024 / dt:10.0f} trial-iterations/s")
            b.close()
