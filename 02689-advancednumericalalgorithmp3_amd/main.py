#!/usr/bin/env python3
"""Launcher of the lid-driven-cavity solvers -- same command line as the reference's Hydra entry.

    python main.py solver=spectral/sg N=64 Re=400                     # one run
    python main.py solver=spectral N=256 Re=1000                      # alias of spectral/sg
    python main.py -m N=64,128,256 Re=100,400,1000                    # grid sweep
    python main.py -m +experiment/validation/saad=spectral            # sweep defined by an experiment file
    python main.py -m +experiment/optimization=corner_smoothing \\
        'solver.corner_smoothing=interval(0.02,0.35)' optuna.objective=botella_vortex

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 main.py -m ...   # one trial per GPU

Contract: reference main.py:75-120 (`run_solver`), :142-225 (objectives), :228-252 (`main`).  Hydra, MLflow
and Optuna are optional: the config tree is composed by `utilities.config.compose`, every run writes
`results.json` (params, metrics, validation errors, Ghia error, objective) and `solution.vts` into its
output directory, and a sweep additionally writes one gathered `sweep_results.json` on rank 0.
"""
from __future__ import annotations

import copy
import json
import logging
import math
import os
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "src"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from utilities.config import compose as C            # noqa: E402
from utilities.sweep.farm import Dist, TPESampler, run_farm  # noqa: E402

log = logging.getLogger("main")


def _jsonable(x):
    if isinstance(x, dict):
        return {k: _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, float) and not math.isfinite(x):
        return repr(x)
    if hasattr(x, "item"):
        return x.item()
    return x


def run_solver(cfg: dict, out_dir: Path, device: str = None) -> dict:
    """Instantiate cfg.solver, solve, validate; returns the result record (reference main.py:75-120)."""
    from solvers import validation as V
    node = dict(cfg["solver"])
    if device is not None:
        node["device"] = device
    solver = C.instantiate(node)
    name = cfg["solver"]["name"]
    n_display = cfg["N"] + 1 if str(name).startswith("spectral") else cfg["N"]
    t0 = time.perf_counter()
    solver.solve()
    errors = solver.compute_validation_errors(reference_dir=cfg.get("validation", {}).get("reference_dir", "data/validation/fv"))
    objective_kind = cfg.get("optuna", {}).get("objective", "fv_l2_error")
    objective = V.compute_optuna_objective(
        objective_kind, errors, solver, cfg["Re"],
        **({"strict_reference_objective": bool(cfg.get("strict_reference_objective", False))}
           if objective_kind == "botella_vortex" else {}))
    rec = dict(run_name=f"{name}_N{n_display}", solver=name, N=cfg["N"], Re=cfg["Re"],
               params=solver.params.to_mlflow(), metrics=solver.metrics.to_mlflow(),
               validation_errors=errors, objective=objective, objective_kind=objective_kind,
               validation_table=solver.validation_table(), total_seconds=time.perf_counter() - t0)
    if int(cfg["Re"]) in V.GHIA_RE:
        rec["ghia"] = solver.ghia_error()
    m = solver.metrics
    rec["steps_per_second"] = m.iterations / m.wall_time_seconds if m.wall_time_seconds > 0 else 0.0
    out_dir.mkdir(parents=True, exist_ok=True)
    (out_dir / "results.json").write_text(json.dumps(_jsonable(rec), indent=1))
    (out_dir / "config.json").write_text(json.dumps(_jsonable({k: v for k, v in cfg.items() if k != "hydra"}), indent=1))
    try:
        solver.to_vtk().save(out_dir / "solution.vts")
    except Exception as exc:                       # a diverged run has NaN fields
        log.warning("VTS export failed: %s", exc)
    _mlflow_log(cfg, rec, solver)
    log.info("Done: %d iter, converged=%s, time=%.2fs", m.iterations, m.converged, m.wall_time_seconds)
    if hasattr(solver, "close"):
        solver.close()
    return rec


def _mlflow_log(cfg, rec, solver):
    """Mirror of the reference's tracking calls, active only when mlflow is installed."""
    try:
        import mlflow
    except ImportError:
        return
    mlflow.set_tracking_uri(cfg.get("mlflow", {}).get("tracking_uri", "./mlruns"))
    mlflow.set_experiment(cfg.get("experiment_name", "LDC-Dev"))
    tags = {"solver": rec["solver"]}
    parent = os.environ.get("MLFLOW_PARENT_RUN_ID")
    if parent:
        tags.update({"mlflow.parentRunId": parent, "parent_run_id": parent, "sweep": "child"})
    with mlflow.start_run(run_name=rec["run_name"], tags=tags, nested=bool(parent)):
        mlflow.log_params(rec["params"])
        mlflow.log_metrics({k: v for k, v in rec["metrics"].items() if isinstance(v, (int, float))})
        if rec["validation_errors"]:
            mlflow.log_metrics(rec["validation_errors"])


def main(argv=None) -> float | None:
    argv = list(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s] %(message)s")
    multirun_flag = False
    conf_dir = HERE / "conf"
    overrides = []
    it = iter(argv)
    for a in it:
        if a in ("-m", "--multirun"):
            multirun_flag = True
        elif a in ("-cd", "--config-dir"):
            conf_dir = Path(next(it))
        else:
            overrides.append(a)

    composer = C.Composer(conf_dir)
    base_cfg, cli_values = composer.compose(overrides)
    multirun = multirun_flag or base_cfg.get("hydra", {}).get("mode") == "MULTIRUN"
    # hydra.mode defaults to MULTIRUN in this tree (as in the reference): sweeps apply with or without -m
    space, fixed = C.sweep_space(base_cfg, cli_values, multirun)
    search = {k: v for k, v in space.items() if isinstance(v, C.Interval)}
    dist = Dist().init()
    device = f"cuda:{dist.local_rank}" if dist.world > 1 else None

    stamp_cfg = C.resolve(C.compose_job(composer, overrides, fixed))
    hy = stamp_cfg.get("hydra", {})
    root_tpl = (hy.get("sweep", {}) if multirun else hy.get("run", {})).get("dir", "hydra_outputs/run")
    root_dir = time.strftime(str(root_tpl).replace("${now:", "").replace("}", ""), time.localtime())
    root_dir = Path(dist.all_gather_object(root_dir)[0])          # every rank uses rank 0's timestamp

    def run_one(assignment, index):
        cfg = C.compose_job(composer, overrides, list(fixed) + list(assignment))
        cfg.setdefault("hydra", {}).setdefault("job", {})["num"] = index
        C.resolve(cfg)
        log.info("Solver: %s, N=%s, Re=%s %s", cfg["solver"]["name"], cfg["N"], cfg["Re"],
                 dict(assignment) if assignment else "")
        rec = run_solver(cfg, root_dir / str(index), device=device)
        rec["overrides"] = {k: v for k, v in assignment}
        return rec

    if not search:
        jobs = C.expand_grid(space)
        trials = [dict(a, N=dict(a).get("N", base_cfg.get("N", 32))) for a in jobs]
        recs = run_farm(trials, lambda t, i: run_one(jobs[i], i), dist)
        objective = recs[0]["objective"] if len(recs) == 1 else None
    else:
        sw = stamp_cfg.get("hydra", {}).get("sweeper", {}) or {}
        n_trials, n_jobs = int(sw.get("n_trials", 15)), max(int(sw.get("n_jobs", 1)), dist.world)
        seed = int((sw.get("sampler") or {}).get("seed", 0))
        sampler = TPESampler(space, seed=seed)
        recs, done = [], 0
        while done < n_trials:
            batch = [sampler.ask() for _ in range(min(n_jobs, n_trials - done))]
            jobs = [list(b.items()) for b in batch]
            out = run_farm([dict(b) for b in batch], lambda t, i: run_one(jobs[i], done + i), dist)
            for b, r in zip(batch, out):
                sampler.tell(b, r["objective"] if isinstance(r["objective"], (int, float)) else math.inf)
            recs.extend(out)
            done += len(batch)
        best, val = sampler.best
        log.info("best trial: %s -> %s", best, val)
        objective = val
    if dist.rank == 0 and (len(recs) > 1 or search):
        root_dir.mkdir(parents=True, exist_ok=True)
        (root_dir / "sweep_results.json").write_text(json.dumps(_jsonable(recs), indent=1))
        log.info("gathered %d trial records -> %s", len(recs), root_dir / "sweep_results.json")
    dist.barrier()
    dist.close()
    return objective


if __name__ == "__main__":
    main()
