#!/usr/bin/env python3
"""Converged solves with the reference's stopping rule (rel. change per step < 1e-6) on the GPU:
iteration counts, wall time, Ghia centreline error (SURVEY 8d Metric 2), Botella table, FV L2 errors.

    python tools/ghia_report.py --cases 64:100,64:400,64:1000,128:1000,256:1000 --out gpurun_out/ghia.json
"""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
    sys.path.insert(0, p)
import __graft_entry__ as g  # noqa: E402

g.build()
from solvers.spectral.sg import SGSolver  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="64:100,64:400,64:1000")
ap.add_argument("--max-iter", type=int, default=4_000_000)
ap.add_argument("--tolerance", type=float, default=1e-6, help="1e-6 = the reference's default (conf/config.yaml:20)")
ap.add_argument("--out", default="gpurun_out/ghia.json")
ap.add_argument("--time-limit", type=float, default=0.0,
                help="seconds per case; when it runs out the state is written to --checkpoint and the case is resumed by "
                     "the next call that finds the file (a gpurun call lasts 20 minutes at most)")
ap.add_argument("--checkpoint", default="", help="npz path of the resumable state (one case)")
a = ap.parse_args()
import threading  # noqa: E402


def heartbeat(stop, label):
    """A long solve prints nothing by itself; gpurun takes 7 silent minutes for a hang."""
    t0 = time.perf_counter()
    while not stop.wait(60.0):
        print(f"  ... {label}: {time.perf_counter() - t0:.0f} s", flush=True)


out = []
if Path(a.out).exists() and (a.tolerance != 1e-6 or a.checkpoint):
    out = json.loads(Path(a.out).read_text())           # append to an earlier (partial) run
for case in a.cases.split(","):
    N, Re = (int(x) for x in case.split(":"))
    s = SGSolver(name="spectral", Re=float(Re), nx=N, ny=N, basis_type="chebyshev", CFL=1.5, beta_squared=5.0,
                 corner_treatment="smoothing", corner_smoothing=0.15, tolerance=a.tolerance, max_iterations=a.max_iter,
                 check_every=4096, graph_iters=32)
    t0 = time.perf_counter()
    stop = threading.Event()
    hb = threading.Thread(target=heartbeat, args=(stop, case), daemon=True)
    hb.start()
    if a.time_limit > 0 and a.checkpoint:
        # resumable form of LidDrivenCavitySolver.solve: same loop, state + control words saved when time runs out
        import numpy as np
        import torch
        from solvers.base import WARMUP_ITERATIONS
        ck = Path(a.checkpoint)
        s._begin(a.tolerance)
        spent, kept_tail = 0.0, None
        if ck.exists():
            z = np.load(ck)
            assert (int(z["N"]), int(z["Re"])) == (N, Re)
            s.set_state(u=z["u"], v=z["v"], p=z["p"])
            s._begin(a.tolerance)                      # T1T / T2T of the restored pressure
            s.d["ctrl"].copy_(torch.from_numpy(z["ctrl"]).to(s.device))
            s.d["scal"].copy_(torch.from_numpy(z["scal"]).to(s.device))
            spent = float(z["spent"])
            print(f"  resumed {case} at iteration {int(z['ctrl'][1])} after {spent:.0f} s", flush=True)
        done, total, last = 0, int(s.d["ctrl"].cpu().numpy()[1]), None
        while total < a.max_iter and not done and time.perf_counter() - t0 < a.time_limit:
            recs, done, total = s._advance(min(4096, a.max_iter - total))
            if len(recs):
                last = recs
        wall = spent + time.perf_counter() - t0
        if not done and total < a.max_iter:
            ck.parent.mkdir(parents=True, exist_ok=True)
            np.savez(ck, N=N, Re=Re, u=s.arrays.u, v=s.arrays.v, p=s.arrays.p, ctrl=s.d["ctrl"].cpu().numpy(),
                     scal=s.d["scal"].cpu().numpy(), spent=wall)
            stop.set()
            print(f"  checkpoint {case}: iteration {total}, rel {last[-1, 0]:.3e}, {wall:.0f} s so far -> {ck}", flush=True)
            s.close()
            continue
        s.history = last
        s._store_results(last[-1:], total, done == 1, wall)
        if ck.exists():
            ck.unlink()
    else:
        s.solve()
    stop.set()
    m = s.metrics
    rec = dict(N=N, Re=Re, tolerance=a.tolerance, kernel_mode=getattr(s, 'kernel_mode', None), iterations=m.iterations, converged=m.converged, wall_time_seconds=m.wall_time_seconds,
               steps_per_second=m.iterations / m.wall_time_seconds, total_seconds=time.perf_counter() - t0,
               final_residual=m.final_residual, psi_min=m.psi_min, psi_min_x=m.psi_min_x, psi_min_y=m.psi_min_y,
               omega_center=m.omega_center, psi_BR=m.psi_BR, psi_BL=m.psi_BL, E=m.final_energy, Z=m.final_enstrophy,
               P=m.final_palinstrophy, u_residual=m.u_momentum_residual, ghia=s.ghia_error(),
               fv=s.compute_validation_errors(), botella=s.validation_table())
    out.append(rec)
    print(json.dumps({k: v for k, v in rec.items() if k != "botella"}), flush=True)
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(out, indent=1))
    s.close()
