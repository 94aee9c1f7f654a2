#!/usr/bin/env python3
"""bench.py -- steady-state pseudo-time-steps/s of the spectral LDC hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   prints ONE JSON line.

* workload  : BASELINE.json configs[2] -- solver=spectral (SG), N=256, Re=1000, fp64,
              CFL=1.5, beta^2=5, cosine lid smoothing 0.15, fluid initially at rest.
* a "step"  : one iteration of LidDrivenCavitySolver.solve (reference base.py:243-313):
              adaptive dt, 4 RK stages with BCs, change norms, residual norms and the
              E/Z/P diagnostics -- everything the reference does per iteration.  The
              step()-only rate (no E/Z/P) is reported beside it as `step_only_value`.
* N > 1     : one process per GPU, each rank advances its own independent trial (the
              sweep axis of the reference; no data-path collective), `value` is the sum
              over ranks of K / max-over-ranks(time): "weak" scaling.  The barrier and the
              gather of the elapsed times are host objects carried by gloo: no RCCL anywhere.
* ghia      : the metric's second half (Ghia centreline error of the bench config) from the committed
              converged-solve reports (profiles/r01_ghia_report.json, r02_ghia_tight.json) -- labelled as a
              committed report, not a live measurement (those solves take 1-56 GPU-minutes).
* roofline  : dominant kernel of the path the solver really takes.  At N=256 that is the chip-wide trial kernel
              (csrc/ldc_wide_kernel.inc, mode 5): ONE launch runs every iteration of a chunk, so a launch's
              algorithmic flops are `iterations_per_launch` x the necessary flops of one full iteration (SURVEY 8d:
              76 M^3 + 2 M Mi (M + Mi)) and `launch_us` is measured with HIP events around launches of
              `iterations_per_launch` = 2048 iterations on the launch stream (the three small launches that close a
              chunk's last record ride along: 0.03 % of it).  `launch_path` holds the same figures for the launch
              path's dominant kernel (the fused RK-stage kernel, 16 M^3 per launch, back-to-back launches), which
              was the headline path until round 3.  `peak` = 78.6 TFLOP/s fp64 matrix (AMD datasheet).  `peak_measured` is the fp64 MFMA rate of a
              micro-benchmark in this run (32 000 back-to-back MFMAs per wave: ~77 TFLOP/s, one
              v_mfma_f64_16x16x4 per 64 cycles and SIMD; the ~48 of rounds 1-3 was the benchmark's own
              code -- AGPR round trips -- not the chip: DESIGN.md section 3).
* cpu_baseline : the NumPy oracle (a port of the reference, pinned to its golden vectors)
              timed on this host's cores for a bounded sample of the same workload.
* timed region: EXACTLY K iterations between barrier + synchronize on both sides, max over ranks -- repeated
              until at least 0.25 s have been timed (K = 20 lasts one millisecond: a single such region measures
              the host's launch latency, not the device); `ms_per_step` is the MEDIAN region / K, `reps` says how
              many regions there were, `launch_path` what the K iterations were enqueued as ("graph": hipGraph
              replays of captures of at most 64 iterations that divide K, "graph+eager" when K has a remainder,
              "persistent": one launch of the persistent trial kernel).
* small_n   : BASELINE.json configs[1] (N=64, Re=400) on the same GPU: full iterations per second through the small-N
              trial kernel (what the solver picks by default at N <= 79) and through the launch path (N = 1 only).
* cu_batch   : 256 trials of N=32 in one batch (N = 1 only): trial-iterations per second through the trial-per-CU kernel.
* --gpus N  : started plainly (no WORLD_SIZE in the environment) with N > 1, the process starts N ranks of itself through
              torch.distributed.run BEFORE it touches the GPU, relays rank 0's JSON line and exits with the ranks' code.
* --headline-only : the headline, its roofline block and nothing else (no farm / small_n / cu_batch / cpu legs): what
              tools/profile_round.py runs under rocprofv3 so that the kernel statistics carry the N=256 launches alone.
* farm      : a second, sweep-shaped measurement for the multi-GPU runs -- every rank advances `trials_per_gpu`
              equal-N trials the way main.py advances the trials a rank owns in the Hydra multirun / Optuna search:
              two batches with shared launches, side by side on two HIP streams of different priority;
              value = trial-iterations/s over all ranks.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
    if p not in sys.path:
        sys.path.insert(0, p)

WORKLOAD = dict(N=256, Re=1000.0)
PEAK_FP64_MFMA_TFLOPS = 78.6     # AMD MI355X datasheet: FP64 matrix (= 128 flop/clk/CU * 256 CU * 2.4 GHz)


def flops_per_step(N: int, diagnostics: bool) -> float:
    """Necessary flops of one SG iteration."""
    M, Mi = N + 1, N - 1
    f = 68.0 * M**3 + 2.0 * M * Mi * (M + Mi)
    if diagnostics:
        f += 8.0 * M**3
    return f


def graph_iters_for(K: int) -> int:
    """Iterations per hipGraph capture so that K iterations are whole replays where possible: the largest
    divisor of K that is <= 64 (K itself when K <= 64); below 8 the captures get too short to amortise a
    replay, then 64 with an eager remainder.  (32 -> 64 iterations per graph: 50.84 -> 50.65 us per iteration at
    N=256, flat beyond: tools/ab_graph_iters.py)"""
    best = max(d for d in range(1, min(K, 64) + 1) if K % d == 0)
    return best if best >= 8 or best == K else 64


def make_solver(N, Re, device, graph_iters=64, persistent=-1):
    from solvers.spectral.sg import SGSolver
    return SGSolver(name="spectral", Re=Re, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N,
                    tolerance=0.0, max_iterations=10**9, basis_type="chebyshev", CFL=1.5,
                    beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
                    multigrid="none", device=device, check_every=8192, graph_iters=graph_iters,
                    persistent=persistent)


def timed_iterations(s, K, diagnostics, barrier):
    """Enqueue exactly K iterations between two host syncs; returns seconds (host clock and events)."""
    import torch
    from solvers.spectral import ldc_lib as L
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    L.check(L.lib().ldc_solver_enqueue(s._handle, K, int(diagnostics), L.stream_ptr()), "enqueue")
    e1.record()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    return t1 - t0, e0.elapsed_time(e1) * 1e-3


def timed_regions(s, K, diagnostics, dist, min_seconds=0.25, max_reps=2000):
    """Regions of exactly K iterations (each bracketed as the contract says, max over ranks) until `min_seconds`
    have been timed; returns (median seconds per region, median event seconds, number of regions)."""
    walls, evs, total = [], [], 0.0
    while (total < min_seconds and len(walls) < max_reps) or len(walls) < 3:
        w, e = timed_iterations(s, K, diagnostics, dist.barrier)
        w = dist.max_float(w)                 # every rank sees the same number, so all leave the loop together
        walls.append(w); evs.append(e)
        total += w
    walls.sort(); evs.sort()
    return walls[len(walls) // 2], evs[len(evs) // 2], len(walls)


def farm_rate(N, B, device, dist, seconds=0.3):
    """Sweep-shaped load: B equal-N trials per rank, advanced the way main.py advances the trials a rank owns -- two
    batches of B/2 with shared launches (solvers.spectral.batched) on two HIP streams, so that the launches of one
    half fill the ramp / drain / hand-over gaps of the other (LDC_BATCH_STREAMS=1: one batch on one stream);
    returns trial-iterations per second of this rank and the iterations timed per trial."""
    import torch
    from solvers.spectral.batched import BatchedSGSolver, run_concurrently
    kw = dict(name="spectral", lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=0.0, max_iterations=10**9,
              basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing", multigrid="none",
              device=device, check_every=4096, graph_iters=64)
    trials = [dict(kw, Re=1000.0, corner_smoothing=0.02 + 0.01 * q) for q in range(B)]
    n_streams = max(1, min(int(os.environ.get("LDC_BATCH_STREAMS", "3")), 2, B))      # one equal-N group: two halves
    cut = [(B * k) // n_streams for k in range(n_streams + 1)]
    halves = [BatchedSGSolver(trials[cut[k]:cut[k + 1]]) for k in range(n_streams)]
    run_concurrently(halves, lambda b: b.run_iterations(64, diagnostics=False), device)      # edge fix, graph build
    K = 512
    state = {"n": 0}

    def advance(b):
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < seconds or n == 0:
            b.run_iterations(K, diagnostics=False)
            n += K
        b.iterations_timed = n

    # five windows, the MEDIAN counts and the spread is reported (a window is 0.3 s of two host threads racing each other:
    # single windows of one box read 105 ... 131 k trial-iterations/s)
    rates = []
    for _ in range(5):
        dist.barrier(); torch.cuda.synchronize()
        dt = run_concurrently(halves, advance, device)
        torch.cuda.synchronize()
        done = sum(len(b) * b.iterations_timed for b in halves)
        rates.append(done / dt)
        state["n"] = min(b.iterations_timed for b in halves)
    for b in halves:
        b.close()
    rates.sort()
    return rates[len(rates) // 2], state["n"], rates[0], rates[-1]


def stage_kernel_time(s, bursts=20, pairs_per_burst=100):
    """Mean duration of one launch of the dominant kernel, stage_kernel<GP=0,LAST=0> (RK stages
    2 and 3): HIP events on the launch stream around bursts of back-to-back launches.  Stage 2
    maps buffer A -> B and stage 3 maps B -> A, so the burst leaves phi^n, dt and p untouched and
    every launch does identical, real work.  The figure includes the inter-launch gap."""
    import torch
    from solvers.spectral import ldc_lib as L
    lib, h, st = L.lib(), s._handle, L.stream_ptr()
    lib.ldc_stage(h, 0, st)                      # fill buffer A from the current state
    times = []
    for _ in range(bursts):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(pairs_per_burst):
            lib.ldc_stage(h, 1, st)
            lib.ldc_stage(h, 2, st)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) * 1e-3 / (2 * pairs_per_burst))
    times.sort()
    return times[len(times) // 2]


def persistent_launch_time(s, K=2048, reps=5):
    """Median duration of one launch of the persistent kernel that runs the loop (mode 3 / 5): HIP events on the launch
    stream around an enqueue of K full iterations (one launch of the trial kernel + the three small launches that close
    the chunk's last record)."""
    import torch
    from solvers.spectral import ldc_lib as L
    lib, h, st = L.lib(), s._handle, L.stream_ptr()
    t = []
    for _ in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        L.check(lib.ldc_solver_enqueue(h, K, 1, st), "enqueue")
        e1.record()
        torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) * 1e-3)
    t = sorted(t[1:])
    return t[len(t) // 2], K


def mfma_peak_measured():
    """fp64 MFMA issue rate of the whole chip (8 independent accumulators per wave)."""
    import torch
    from solvers.spectral import ldc_lib as L
    grid, iters = 256 * 8, 4000
    sink = torch.zeros(grid * 256, dtype=torch.float64, device="cuda")
    lib, st = L.lib(), L.stream_ptr()
    lib.ldc_mfma_peak(sink.data_ptr(), 100, grid, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    lib.ldc_mfma_peak(sink.data_ptr(), iters, grid, st)
    e1.record()
    torch.cuda.synchronize()
    flops = grid * 4 * iters * 8 * 2048.0
    return flops / (e0.elapsed_time(e1) * 1e-3) / 1e12


def _cpu_budget():
    """CPUs this process may really use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max") and txt[0] != "max":
                n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            elif f.endswith("quota_us") and int(txt[0]) > 0:
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                n = min(n, max(1, int(int(txt[0]) / per)))
        except Exception:
            pass
    return n


def cpu_baseline(N, Re, budget_s=14.0):
    """The oracle's full solve() iteration timed on the host cores (bounded sample).  A short
    probe picks the OpenBLAS thread count that is fastest on this box (oversubscribed pools
    are much slower), so the CPU gets its best shot."""
    import numpy as np
    from oracle.ldc_oracle import OracleSG
    from threadpoolctl import threadpool_limits
    o = OracleSG(N, Re)
    up, vp = o.u.copy(), o.v.copy()

    def one():
        nonlocal up, vp
        o.step()
        np.linalg.norm(o.u - up); np.linalg.norm(o.v - vp); np.linalg.norm(up); np.linalg.norm(vp)
        o.residual_norms(); o.energy(); o.enstrophy(); o.palinstrophy()
        up, vp = o.u.copy(), o.v.copy()

    cap = _cpu_budget()
    cands = sorted({max(1, c) for c in (cap, cap // 2, cap // 4, 8, 4) if c <= cap}, reverse=True)
    best, best_rate = cands[0], 0.0
    for c in cands:
        with threadpool_limits(limits=c, user_api="blas"):
            one()
            t0 = time.perf_counter()
            for _ in range(6):
                one()
            rate = 6 / (time.perf_counter() - t0)
        if rate > best_rate:
            best, best_rate = c, rate
    with threadpool_limits(limits=1, user_api="blas"):
        one()
        n1, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            one()
            n1 += 1
        rate1 = n1 / (time.perf_counter() - t0)
    with threadpool_limits(limits=best, user_api="blas"):
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < budget_s - 3.0:
            one()
            n += 1
        dt = time.perf_counter() - t0
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return dict(value=n / dt, unit="steps/s", cores=int(best), kind="port", value_1thread=rate1, cpu_model=model,
                cpus_available=int(cap),
                sample=f"{n} full solve() iterations (step + norms + E/Z/P) of the NumPy oracle at "
                       f"N={N}, Re={Re:g} from rest, {dt:.1f} s, OpenBLAS {best} threads "
                       f"(best of {cands}; {cap} CPUs available)")


def _plain_stage_entry(kernels: dict):
    """The dominant kernel's entry of a PMC summary: stage_kernel<GP=0, LAST=0, DUMP=0, BATCH=0, DIAG=0[, RECT=0]> (the square-grid
    instantiation; the last template argument exists since nx != ny is supported)."""
    for name in ("stage_kernel<false, false, false, false, 0, false>", "stage_kernel<false, false, false, false, 0>"):
        if name in kernels:
            return kernels[name]
    return None


def _wide_entry(kernels: dict):
    """The chip-wide kernel's entry of a PMC summary (SG with diagnostics, tail layout: the N=256 headline)."""
    for name, v in kernels.items():
        if name.startswith("wide_kernel<false, true, true>"):
            return v
    return None


def pmc_traffic(N, wide=False):
    """HBM/fabric bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/r*_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of tools/pmc_run.py at
    N=256, gfx950 correction applied as MI355X_MICROARCH.md prescribes).  None for other sizes, and for the chip-wide kernel
    until a pass of it is committed (the figure is then per launch of `iterations_per_launch` iterations)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if N != 256 or not files:
        return None, None, None
    for f in reversed(files):
        with open(f) as fh:
            d = json.load(fh)
        k = _wide_entry(d["kernels"]) if wide else _plain_stage_entry(d["kernels"])
        if k and "hbm_bytes_per_launch" in k:
            return k["hbm_bytes_per_launch"], os.path.relpath(f, ROOT), k.get("iterations_per_launch")
    return None, None, None


def pmc_mfma_util(N, launch_seconds):
    """MFMA utilisation of the dominant kernel from the committed SQ-counter pass (profiles/r*_pmc.json, written by
    tools/pmc_summarize.py): SQ_VALU_MFMA_BUSY_CYCLES per launch (busy cycles of the matrix pipes summed over the
    SIMDs: 64 per fp64 MFMA) over the SIMD-cycles of a launch -- 1024 SIMDs at the 2.4 GHz peak clock -- for the launch
    time measured live in this run; `mfma_util_profiled` is the same quotient with the (longer) launch time of the
    profiled pass itself.  None when no such pass is committed for this size."""
    import glob
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))]
    if N != 256 or not files:
        return None
    for f in reversed(files):
        with open(f) as fh:
            k = _plain_stage_entry(json.load(fh)["kernels"]) or {}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in k:
            busy = k["SQ_VALU_MFMA_BUSY_CYCLES"]
            return {"mfma_util": busy / (launch_seconds * 2.4e9 * 1024), "mfma_util_profiled": k.get("mfma_util"),
                    "mfma_busy_cycles_per_launch": busy, "mfma_util_source": os.path.relpath(f, ROOT),
                    "sq_wait_inst_frac": k.get("sq_wait_inst_frac"), "sq_wait_any_frac": k.get("sq_wait_any_frac")}
    return None


def ghia_block(N, Re):
    """Second half of BASELINE's metric ("...; Ghia centreline L2 error", SURVEY 8d Metric 2) for the bench config.  NOT
    measured in this run: the converged solves take 70 s (reference stopping rule, 1.3 M iterations) and 56 GPU-minutes
    (iterated to 1e-9, 64.6 M iterations); the figures are read from the committed reports of those solves
    (tools/ghia_report.py) and labelled as such."""
    def pick(path, tol):
        f = ROOT / "profiles" / path
        if not f.exists():
            return None
        for r in json.loads(f.read_text()):
            if int(r["N"]) == int(N) and float(r["Re"]) == float(Re) and float(r["tolerance"]) == tol:
                g = r["ghia"]
                return {"u_rms": g["u_rms"], "v_rms": g["v_rms"], "u_rel": g["u_rel"], "v_rel": g["v_rel"],
                        "iterations": r["iterations"], "psi_min": r.get("psi_min"), "source": f"profiles/{path}"}
        return None
    ref_rule = pick("r01_ghia_report.json", 1e-6)
    tight = pick("r02_ghia_tight.json", 1e-9)
    if ref_rule is None and tight is None:
        return None
    return {"kind": "committed report, not a live measurement",
            "metric": "RMS over Ghia et al. (1982) centreline table points of (interpolated solver profile - table)",
            "reference_stopping_rule_tol_1e-6": ref_rule,
            "converged_tol_1e-9": tight,
            "note": "with the reference's own stopping rule (relative change per step < 1e-6) the solve stops ~2 % of "
                    "the way to steady state at this N (dt ~ dx_min^2): the reference would report the same numbers "
                    "(its trajectory is reproduced to 1e-12); the converged figures are the meaningful ones"}


def small_n_block(device, N=64, Re=400.0, K=4096):
    """BASELINE.json configs[1] (solver=spectral N=64 Re=400 on one MI355X) beside the headline: full solve() iterations per
    second as the solver runs that size by default -- the small-N trial kernel (csrc/ldc_xcd_kernel.inc: all iterations of a
    chunk in one launch, the trial's 5 x 5 work-groups on one XCD) -- and on the launch path (persistent=0) for comparison.
    Median of five regions of K iterations each, HIP events on the launch stream."""
    import torch
    from solvers.spectral import ldc_lib as L
    out = {"workload": f"solver=spectral (SG) N={N} Re={Re:g} fp64, full solve() iteration, fluid from rest", "steps": K}
    for name, mode in (("value", -1), ("launch_path_value", 0)):
        s = make_solver(N, Re, device, graph_iters=64, persistent=mode)
        s._begin(0.0)
        resolved = int(L.lib().ldc_solver_mode(s._handle))
        t = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            L.check(L.lib().ldc_solver_enqueue(s._handle, K, 1, L.stream_ptr()), "enqueue")
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) * 1e-3)
        assert L.lib().ldc_solver_status(s._handle) == 0, "small-N trial kernel gave up a wait (LDC_E_SYNC)"
        rec = s.d["rec"].cpu().numpy()
        assert bool((rec == rec).all()), "non-finite history record at N=64"
        s.close()
        t = sorted(t[1:])
        out[name] = K / t[len(t) // 2]
        if mode == -1:
            out["mode"] = resolved
            out["us_per_step"] = 1e6 * t[len(t) // 2] / K
    out["unit"] = "steps/s"
    return out



def cu_batch_block(device, N=32, B=256, K=2048):
    """The other end of the sweep axis on one GPU: MANY small trials (the reference's Optuna study samples N = 30, 40, 50;
    its fixtures are N = 16, 32).  B trials of size N in one batch: from LDC_CU_AUTO_TRIALS trials on the library advances
    them with the trial-per-CU kernel (csrc/ldc_cu_kernel.inc: one work-group per trial, stage state in LDS, no hand-over
    between work-groups); trial-iterations per second over a chunk of K iterations (step()-only loop, host clock around
    enqueue + wait + the batch-wide copy of the history rows: what a sweep pays per chunk)."""
    import torch
    from solvers.spectral import ldc_lib as L
    from solvers.spectral.batched import BatchedSGSolver
    # (Re = 100 ... 228: at N = 32 the reference's scheme itself diverges at Re = 1000 after ~2 300 iterations -- the oracle too)
    trials = [dict(name="spectral", Re=100.0 + 0.5 * q, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N, tolerance=0.0,
                   max_iterations=10**9, basis_type="chebyshev", CFL=1.5, beta_squared=5.0, corner_treatment="smoothing",
                   corner_smoothing=0.02 + 0.0005 * q, multigrid="none", device=device, check_every=K, graph_iters=64)
              for q in range(B)]
    b = BatchedSGSolver(trials)
    b.run_iterations(64, diagnostics=False)
    mode = int(L.lib().ldc_batch_mode(b._batch))
    rates = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b.run_iterations(K, diagnostics=False)
        torch.cuda.synchronize()
        rates.append(B * K / (time.perf_counter() - t0))
    rec = b.solvers[B - 1].d["rec"].cpu().numpy()
    assert bool((rec == rec).all()), "non-finite history record in the batch of small trials"
    b.close()
    rates.sort()
    return {"value": rates[len(rates) // 2], "statistic": "median of 5 chunks", "min": rates[0], "max": rates[-1],
            "unit": "trial-iterations/s", "N": N, "trials": B, "iterations_per_chunk": K, "batch_mode": mode,
            "workload": f"{B} SG trials of N={N} in one batch on one GPU (step()-only loop), one work-group per trial"}



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--N", type=int, default=WORKLOAD["N"])
    ap.add_argument("--Re", type=float, default=WORKLOAD["Re"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-farm", action="store_true", help="skip the batched-trials (sweep-shaped) measurement")
    ap.add_argument("--headline-only", action="store_true",
                    help="the headline and its roofline block only (what tools/profile_round.py profiles)")
    ap.add_argument("--leg", default="", choices=["", "farm", "small_n", "cu_batch"],
                    help="ONE secondary leg only (no headline): what tools/profile_round.sh profiles leg by leg")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU work: the launcher, the rendezvous, the barrier, the max-over-ranks and the gather only (CPU test of --gpus N)")
    ap.add_argument("--persistent", type=int, default=-1,
                    help="-1 the library's choice (N=256: the chip-wide kernel), 0 launch per stage, 3 / 5 the persistent kernels")
    a = ap.parse_args()
    if a.headline_only:
        a.no_cpu = a.no_farm = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started plainly: become the launcher.  N fresh ranks of this script through torch.distributed.run, BEFORE anything
        # here has touched the GPU (a process that has initialised HIP must never be replaced or forked into ranks); rank
        # 0's JSON line is relayed, the exit code is the ranks'.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
        if lines:
            print(lines[-1])
        else:
            sys.stdout.write(r.stdout)
        raise SystemExit(r.returncode if r.returncode != 0 or lines else 1)

    import torch
    from utilities.sweep.farm import Dist
    dist = Dist()
    rank, world, local = dist.rank, dist.world, dist.local_rank
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start bench.py plainly (it launches its own ranks) or with {a.gpus} ranks")
    if a.dry_run:
        # the multi-rank flow without a device: what tests/test_bench_cpu.py runs through the plain command on CPU
        dist.init("gloo")
        dist.barrier()
        t = dist.max_float(1e-3 * (1 + rank))
        got = dist.all_gather_object(rank)
        dist.barrier()
        dist.close()
        if rank == 0:
            print(json.dumps({"metric": "steady-state time-steps/sec at N=256 Re=1000", "dry_run": True, "value": None,
                              "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ranks_seen": got, "max_over_ranks_s": t}))
        return
    # (LDC_DIST_BACKEND=gloo + fewer cards than ranks: a rehearsal of the multi-rank flow on a one-GPU box; the ranks
    #  then share the card and the rates mean nothing)
    shared_card = world > max(1, torch.cuda.device_count())
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dist.local_rank = local
    if shared_card and a.persistent != 0:
        # ranks that share a card cannot both have a work-group on every CU: the persistent kernels (whose work-groups wait
        # for each other) would each hold part of the chip until their bounded waits give up -- launch path for the rehearsal
        a.persistent = 0
    # The ranks never exchange device data: the barrier and the max-over-ranks of the elapsed time are host objects, and
    # gloo carries those (as main.py's farm does under LDC_DIST_BACKEND=gloo) -- the replicas-only design needs no RCCL.
    dist.init(os.environ.get("LDC_DIST_BACKEND", "gloo"))
    barrier = dist.barrier

    import __graft_entry__ as g
    if rank == 0:
        g.build()
    barrier()
    g._paths()

    if a.leg:
        # one secondary leg alone (so that a rocprofv3 pass over this command holds that leg's launches only)
        if a.leg == "farm":
            rate, n_it, r_lo, r_hi = farm_rate(128, 8, f"cuda:{local}", dist)
            leg = {"value": rate, "min": r_lo, "max": r_hi, "unit": "trial-iterations/s", "N": 128, "trials_per_gpu": 8}
        elif a.leg == "small_n":
            leg = small_n_block(f"cuda:{local}")
        else:
            leg = cu_batch_block(f"cuda:{local}")
        dist.close()
        if rank == 0:
            print(json.dumps({"leg": a.leg, a.leg: leg}))
        return

    gi = int(os.environ.get("LDC_BENCH_GRAPH_ITERS", graph_iters_for(a.steps)))       # (the override is for experiments)
    s = make_solver(a.N, a.Re, f"cuda:{local}", graph_iters=gi, persistent=a.persistent)
    s._begin(0.0)
    from solvers.spectral import ldc_lib as L
    mode = int(L.lib().ldc_solver_mode(s._handle))         # what the library resolved: 0 launch per stage, 3 small-N kernel
    launch_path = ({3: "persistent (small-N kernel)", 4: "persistent (trial-per-CU kernel)", 5: "persistent (chip-wide kernel)"}.get(mode, "persistent")
                   if mode != 0 else ("graph" if a.steps % gi == 0 else "graph+eager"))
    # warm-up (also instantiates both hipGraphs): W untimed steps, at least two full captures
    timed_iterations(s, max(a.warmup, 2 * gi), True, barrier)
    timed_iterations(s, 2 * gi, False, barrier)

    wall, ev, reps = timed_regions(s, a.steps, True, dist)
    wall_so, _, _ = timed_regions(s, a.steps, False, dist)
    ctrl = s.d["ctrl"].cpu().numpy()
    assert int(ctrl[0]) == 0, "latch fired during the bench (tolerance is 0: must not happen)"
    rec = s.d["rec"].cpu().numpy()
    assert bool((rec == rec).all()), "non-finite history record: the timed run diverged"
    if mode != 0:     # a persistent launch that gave up a barrier wait leaves undefined state, not a rate
        assert L.lib().ldc_solver_status(s._handle) == 0, "persistent kernel gave up a barrier wait (LDC_E_SYNC)"

    farm = None
    if not a.no_farm:
        fN, fB = 128, 8                       # config 5's shape: N = 128, 64 trials over 8 GPUs = 8 per GPU and round
        rate, n_it, r_lo, r_hi = farm_rate(fN, fB, f"cuda:{local}", dist)
        rates = dist.all_gather_object(rate)
        farm = {"value": float(sum(rates)), "unit": "trial-iterations/s", "n_gpus": world, "trials_per_gpu": fB, "N": fN,
                "per_gpu": [float(r) for r in rates], "iterations_timed_per_trial": n_it,
                "windows": 5, "statistic": "median of the windows", "rank0_min": float(r_lo), "rank0_max": float(r_hi),
                "streams": max(1, min(int(os.environ.get("LDC_BATCH_STREAMS", "3")), 2, fB)),
                "workload": f"{fB} SG trials of N={fN} per GPU as main.py advances the trials a rank owns: two batches with "
                            "shared launches on two HIP streams (step()-only loop)",
                "meaning": "throughput of INDEPENDENT trials (a grid sweep, or one round of a search): this is what scales with "
                           "the number of GPUs.  A model-based search scales by rounds(1 GPU) / rounds(N GPUs) only: the sampler "
                           "learns between rounds, main.py never plans fewer than three (search_mode=throughput; config 5: 8 -> 3 "
                           "rounds, at most 2.7x) and offers the reference's own ask-n_jobs / tell-n_jobs sequence as "
                           "search_mode=reference (no scaling beyond batching) -- DESIGN.md 3"}

    out = None
    if rank == 0:
        M = a.N + 1
        peak_meas = mfma_peak_measured()
        f_iter = flops_per_step(a.N, True)
        if mode != 0:
            # the persistent kernel that runs the loop: one launch = every iteration of a chunk
            t_launch, k_launch = persistent_launch_time(s)
            f_launch = k_launch * f_iter
            achieved = f_launch / t_launch / 1e12
            traffic, traffic_src, traffic_iters = pmc_traffic(a.N, wide=True)
            if traffic is not None and traffic_iters:
                traffic = traffic / traffic_iters * k_launch        # the PMC pass's launch holds `traffic_iters` iterations
            kname = {3: "xcd_kernel (small-N trial kernel: all iterations of a chunk in one launch, one XCD)",
                     4: "cu_kernel (trial-per-CU kernel)",
                     5: "wide_kernel<SG, E/Z/P" + (", tail layout" if (a.N % 16 == 0) else "") + "> (chip-wide trial kernel: all "
                        "iterations of a chunk in one launch, one work-group per CU)"}.get(mode, f"mode {mode}")
            roof = {"bound": "mfma", "kernel": kname, "mode": mode, "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                    "traffic": traffic, "traffic_unit": "bytes/launch (FETCH_SIZE+WRITE_SIZE)", "traffic_source": traffic_src,
                    "traffic_basis": (f"PMC pass of a launch of {traffic_iters} iterations, scaled to iterations_per_launch"
                                      if traffic_iters else None),
                    "flops_per_launch": f_launch, "launch_us": t_launch * 1e6, "iterations_per_launch": k_launch,
                    "flops_per_iteration": f_iter, "us_per_iteration_in_launch": t_launch * 1e6 / k_launch,
                    "peak_measured": peak_meas, "frac_of_measured": achieved / peak_meas}
            # the launch path's dominant kernel beside it (the headline path until round 3): same solver state, mode 0
            s0 = make_solver(a.N, a.Re, f"cuda:{local}", graph_iters=gi, persistent=0)
            s0._begin(0.0)
            timed_iterations(s0, 2 * gi, True, lambda: None)
            t_stage = stage_kernel_time(s0)
            w0, _, _ = timed_regions(s0, a.steps, True, dist) if world == 1 else (None, None, None)
            s0.close()
            a_stage = 16.0 * M**3 / t_stage / 1e12
            roof["launch_path"] = {"kernel": "stage_kernel<GP=0,LAST=0,DUMP=0> (fused RK stage, one launch per stage)",
                                   "flops_per_launch": 16.0 * M**3, "launch_us": t_stage * 1e6, "achieved": a_stage,
                                   "frac": a_stage / PEAK_FP64_MFMA_TFLOPS,
                                   "value": (a.steps / w0) if w0 else None, "ms_per_step": (1e3 * w0 / a.steps) if w0 else None}
        else:
            t_stage = stage_kernel_time(s)
            f_launch = 16.0 * M**3          # 8 contractions x 2 M^3 (stage 1 adds 4 M^3 for grad p)
            achieved = f_launch / t_stage / 1e12
            traffic, traffic_src, _ = pmc_traffic(a.N)
            roof = {"bound": "mfma", "kernel": "stage_kernel<GP=0,LAST=0,DUMP=0> (fused RK stage)", "mode": 0, "achieved": achieved,
                    "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP64_MFMA_TFLOPS,
                    "traffic": traffic, "traffic_unit": "bytes/launch (FETCH_SIZE+WRITE_SIZE)",
                    "traffic_source": traffic_src, "flops_per_launch": f_launch, "launch_us": t_stage * 1e6,
                    "peak_measured": peak_meas, "frac_of_measured": achieved / peak_meas}
            mfma = pmc_mfma_util(a.N, t_stage)
            if mfma is not None:
                roof.update(mfma)
        # the WHOLE iteration as the driver times it against the same peak: necessary flops of one full solve() iteration
        # (76 M^3 + 2 M Mi (M + Mi)) over ms_per_step -- launch gaps and per-chunk costs included
        roof["iteration_frac"] = f_iter * a.steps / wall / 1e12 / PEAK_FP64_MFMA_TFLOPS
        roof["iteration_flops"] = f_iter
        out = {
            "metric": "steady-state time-steps/sec at N=256 Re=1000",
            "value": world * a.steps / wall, "unit": "steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": 1e3 * wall / a.steps, "reps": reps, "launch_path": launch_path,
            "solver_mode": mode, "graph_iters": gi, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"solver=spectral (SG) N={a.N} Re={a.Re:g} fp64, full solve() iteration "
                                   "(dt, 4 RK stages + BCs, change/residual norms, E/Z/P), fluid from rest",
                       "N": a.N, "Re": a.Re, "CFL": 1.5, "beta_squared": 5.0, "corner_smoothing": 0.15,
                       "trials_per_gpu": 1, "parallelism": f"{world} independent trial(s), one per GPU"},
            "step_only_value": world * a.steps / wall_so, "step_only_ms": 1e3 * wall_so / a.steps,
            "event_ms_per_step": 1e3 * ev / a.steps,
            "iteration_tflops": f_iter * a.steps / wall / 1e12,
            "ghia": ghia_block(a.N, a.Re),
            "roofline": roof,
        }
        out["farm"] = farm
        out["small_n"] = small_n_block(f"cuda:{local}") if (world == 1 and not a.headline_only) else None
        out["cu_batch"] = cu_batch_block(f"cuda:{local}") if (world == 1 and not a.headline_only) else None
        if world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_baseline(a.N, a.Re)
            out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
    s.close()
    barrier()
    dist.close()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
