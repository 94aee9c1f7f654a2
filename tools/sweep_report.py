#!/usr/bin/env python3
"""BASELINE configs 4 and 5 at FULL size through main.py on one GPU; writes a markdown table (development aid).

    python tools/sweep_report.py config4 gpurun_out/sweeps/config4.md
    python tools/sweep_report.py config5 gpurun_out/sweeps/config5.md [n_trials] [n_jobs]
    python tools/sweep_report.py optuna_ref gpurun_out/sweeps/optuna_ref.md [n_trials] [n_jobs]
(optuna_ref: the reference's experiment file as it stands -- conf/experiment/optimization/corner_smoothing.yaml: FSG, Re=1000,
N sampled from {30, 40, 50}, 15 trials in rounds of 5 -- or with more trials per round)
"""
import importlib.util
import json
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "02689-advancednumericalalgorithmp3_amd"
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

g.build()
spec = importlib.util.spec_from_file_location("ldc_main_report", PKG / "main.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)

which, out = sys.argv[1], Path(sys.argv[2]).resolve()
out.parent.mkdir(parents=True, exist_ok=True)
work = out.parent / f"{which}_work"
work.mkdir(exist_ok=True)
import os  # noqa: E402

os.chdir(work)
stop = threading.Event()


def heartbeat():
    t0 = time.perf_counter()
    while not stop.wait(60.0):
        print(f"  ... {which}: {time.perf_counter() - t0:.0f} s", flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
t0 = time.perf_counter()
if which == "config4":
    argv = ["-m", "N=64,128,256", "Re=100,400,1000"]
elif which == "optuna_ref":
    argv = ["-m", "+experiment/optimization=corner_smoothing", "optuna.objective=botella_vortex"]
    if len(sys.argv) > 3:
        argv.append(f"hydra.sweeper.n_trials={sys.argv[3]}")
    if len(sys.argv) > 4:
        argv.append(f"hydra.sweeper.n_jobs={sys.argv[4]}")
else:
    n_trials = sys.argv[3] if len(sys.argv) > 3 else "64"
    n_jobs = sys.argv[4] if len(sys.argv) > 4 else "8"
    argv = ["-m", "+experiment/optimization=corner_smoothing", "N=128", f"hydra.sweeper.n_trials={n_trials}",
            f"hydra.sweeper.n_jobs={n_jobs}", "optuna.objective=botella_vortex"]
best = mod.main(argv)
wall = time.perf_counter() - t0
stop.set()
root = sorted(work.glob("hydra_outputs/multirun/*/*"))[-1]
recs = json.loads((root / "sweep_results.json").read_text())
lines = [f"`main.py {' '.join(argv)}` on ONE MI355X: {len(recs)} trials, {wall:.0f} s end to end", ""]
if which == "config4":
    lines += ["| N | Re | iterations | converged | psi_min | Ghia u_rms / v_rms | FV u_L2 | batch size | batch wall s |",
              "|---|---|---|---|---|---|---|---|---|"]
    for r in recs:
        m, gh = r["metrics"], r.get("ghia", {})
        lines.append(f"| {r['N']} | {r['Re']} | {m['iterations']} | {m['converged']} | {m['psi_min']:.6f} | "
                     f"{gh.get('u_rms', float('nan')):.4f} / {gh.get('v_rms', float('nan')):.4f} | "
                     f"{r['validation_errors'].get('u_L2_error', float('nan')):.4f} | "
                     f"{r.get('solve_batch_size', r['batch_size'])} | {r.get('solve_batch_seconds', r['batch_seconds']):.1f} |")
    its = sum(r["metrics"]["iterations"] for r in recs)
    # the rank's trials are ONE farm group whose batches overlap on the worker streams: its wall time is the total
    secs = max(r["batch_seconds"] for r in recs)
    lines += ["", f"Total: {its} trial-iterations in {secs:.0f} s of solver wall time "
                  f"({recs[0].get('solve_streams', 1)} worker stream(s); a batch's own wall time overlaps the others') = "
                  f"{its / secs:.0f} trial-iterations/s."]
else:
    lines += ["| trial | N | corner_smoothing | iterations | converged | objective | psi_min | x | y | batch wall s |",
              "|---|---|---|---|---|---|---|---|---|---|"]
    for k, r in enumerate(recs):
        if "metrics" not in r:
            lines.append(f"| {k} | {r.get('N', '')} | | | | failed: {str(r.get('error', ''))[:60]} | | | | |")
            continue
        m = r["metrics"]
        lines.append(f"| {k} | {r['N']} | {r['params']['corner_smoothing']:.4f} | {m['iterations']} | {m['converged']} | {r['objective']:.5f} | "
                     f"{m['psi_min']:.6f} | {m['psi_min_x']:.4f} | {m['psi_min_y']:.4f} | {r.get('solve_pool_seconds', r['batch_seconds']):.1f} |")
    its = sum(r["metrics"]["iterations"] for r in recs if "metrics" in r)
    lines += ["", f"Best objective {best:.5f}; {its} trial-iterations, {its / wall:.0f} trial-iterations/s end to end."]
out.write_text("\n".join(lines) + "\n")
print("\n".join(lines[:6]))
import shutil  # noqa: E402

os.chdir(out.parent)
shutil.rmtree(work, ignore_errors=True)       # the run directories (one solution.vts per trial) do not travel back
