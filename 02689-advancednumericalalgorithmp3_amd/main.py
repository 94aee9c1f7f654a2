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


def make_record(cfg: dict, solver, out_dir: Path, t0: float) -> dict:
    """Everything the reference does after solve() (main.py:99-119): validation, objective, artefacts."""
    from solvers import validation as V
    name = cfg["solver"]["name"]
    n_display = cfg["N"] + 1 if str(name).startswith("spectral") else cfg["N"]
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
    return rec


def run_solver(cfg: dict, out_dir: Path, device: str = None) -> dict:
    """Instantiate cfg.solver, solve, validate; returns the result record (reference main.py:75-120)."""
    node = dict(cfg["solver"])
    if device is not None:
        node["device"] = device
    solver = C.instantiate(node)
    t0 = time.perf_counter()
    solver.solve()
    rec = make_record(cfg, solver, out_dir, t0)
    if hasattr(solver, "close"):
        solver.close()
    return rec


def run_batch(cfgs: list, out_dirs: list, device: str = None) -> list:
    """Several SG (or several FSG) trials of equal N on one GPU, advanced by the same launches
    (solvers.spectral.batched)."""
    from solvers.spectral.batched import BatchedFSGSolver, BatchedSGSolver
    nodes = []
    for cfg in cfgs:
        node = {k: v for k, v in cfg["solver"].items() if k != "_target_"}
        if device is not None:
            node["device"] = device
        nodes.append(node)
    t0 = time.perf_counter()
    fsg = cfgs[0]["solver"]["_target_"].endswith("FSGSolver")
    batch = (BatchedFSGSolver if fsg else BatchedSGSolver)(nodes)
    batch.solve()
    recs = [make_record(cfg, s, d, t0) for cfg, s, d in zip(cfgs, batch.solvers, out_dirs)]
    batch.close()
    return recs


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

    SG, FSG = "solvers.spectral.sg.SGSolver", "solvers.spectral.fsg.FSGSolver"
    max_batch = int(os.environ.get("LDC_MAX_BATCH", stamp_cfg.get("hydra", {}).get("launcher", {}).get("batch_trials", 64)))

    def job_cfg(assignment, index):
        cfg = C.compose_job(composer, overrides, list(fixed) + list(assignment))
        cfg.setdefault("hydra", {}).setdefault("job", {})["num"] = index
        return C.resolve(cfg)

    def run_group(items, jobs, offset=0):
        """items: [(index, trial)] owned by this rank with one group key; SG trials of equal N share launches."""
        cfgs = [job_cfg(jobs[i], offset + i) for i, _ in items]
        out = []
        targets = {c["solver"]["_target_"] for c in cfgs}
        levels = {int(c["solver"].get("n_levels", 0)) for c in cfgs}
        if targets in ({SG}, {FSG}) and len(levels) == 1 and len(cfgs) > 1 and max_batch > 1:
            for lo in range(0, len(cfgs), max_batch):
                part = cfgs[lo: lo + max_batch]
                log.info("batch of %d trials at N=%s on %s", len(part), part[0]["N"], device or "cuda:0")
                recs = run_batch(part, [root_dir / str(offset + i) for i, _ in items[lo: lo + max_batch]], device)
                out.extend(recs)
        else:
            for (i, _), cfg in zip(items, cfgs):
                out.append(run_solver(cfg, root_dir / str(offset + i), device=device))
        for (i, _), r in zip(items, out):
            r["overrides"] = {k: v for k, v in jobs[i]}
        return out

    def key_of(jobs):
        return lambda t: (t.get("N"), str(dict(jobs[t["_job"]]).get("solver", "")))

    if not search:
        jobs = C.expand_grid(space)
        trials = [dict(a, N=dict(a).get("N", base_cfg.get("N", 32)), _job=i) for i, a in enumerate(jobs)]
        recs = run_farm(trials, None, dist, run_group=lambda items: run_group(items, jobs), group_key=key_of(jobs))
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
            trials = [dict(b, N=b.get("N", base_cfg.get("N", 32)), _job=i) for i, b in enumerate(batch)]
            out = run_farm(trials, None, dist, run_group=lambda items: run_group(items, jobs, done),
                           group_key=key_of(jobs))
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
