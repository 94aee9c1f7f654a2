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
from utilities.sweep.farm import Dist, FarmError, TPESampler, plan_rounds, run_farm, trial_cost  # noqa: E402

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
    if hasattr(solver, "kernel_mode"):          # which of the library's kernels advanced this trial (include/ldc_hip.h: 0, 3, 4, 5)
        rec["kernel_mode"] = int(solver.kernel_mode)
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
    mlflow = _mlflow()
    if mlflow is None:
        solver.solve()
        rec = make_record(cfg, solver, out_dir, t0)
    else:
        from utilities.tracking import sweep as T
        if _TRACKER is None:
            mlflow.set_tracking_uri(cfg.get("mlflow", {}).get("tracking_uri", "./mlruns"))
            mlflow.set_experiment(T.experiment_name(cfg))
        name = cfg["solver"]["name"]
        n_display = cfg["N"] + 1 if str(name).startswith("spectral") else cfg["N"]
        with T.open_child_run(mlflow, name, f"{name}_N{n_display}", _parent_of(cfg)):
            solver.solve()                          # live metrics every 50 iterations go to this run
            rec = make_record(cfg, solver, out_dir, t0)
    if hasattr(solver, "close"):
        solver.close()
    return rec


def _device_cus(device: str = None) -> int:
    """Compute units of the device the batches will run on, as the LIBRARY counts them (ldc_device_info: the same number
    its co-resident kernels size their launches by; nothing is asked of torch's device bookkeeping)."""
    import ctypes
    import torch
    from solvers.spectral import ldc_lib as L
    cus, xcds = ctypes.c_int(), ctypes.c_int()
    with torch.cuda.device(torch.device(device if device is not None else "cuda")):
        L.check(L.lib().ldc_device_info(ctypes.byref(cus), ctypes.byref(xcds)), "ldc_device_info")
    return int(cus.value)


def run_batches(groups: list, device: str = None) -> list:
    """groups: [(cfgs, out_dirs)], each a set of SG (or FSG) trials of equal N that can share their launches.
    Returns the record lists in the same order.

    Every group is cut into two batches (solvers.spectral.batched); the batches of ALL groups then go, longest first,
    through a pool of LDC_BATCH_STREAMS host threads (default 3: one per stream priority of the hardware), one HIP
    stream each.  The launches of the batch on one stream fill the ramp / drain / hand-over gaps of the
    batch on the other -- within one N (two halves of a batch: up to 1.3x) and across sizes (the small-N groups of a
    grid run in the shadow of the large ones).  A batch that fails leaves error records for its trials only.
    Records (validation, artefacts, MLflow) are made afterwards, in the caller's thread."""
    import threading
    from solvers.spectral.batched import BatchedFSGSolver, BatchedSGSolver, run_concurrently
    n_workers = max(1, int(os.environ.get("LDC_BATCH_STREAMS", "3")))
    n_cus = _device_cus(device)      # decides whether a size fills the chip on its own (N >= 241 on 256 CUs)
    tasks = []
    for gi, (cfgs, _) in enumerate(groups):
        parts = max(1, min(2, n_workers, len(cfgs)))      # halves: more, smaller batches measured no better (N=128)
        # a trial whose tiles fill the chip on their own (N >= 241 on 256 CUs) gains nothing from sharing launches (two
        # N=256 trials batched: 20.4 k trial-iterations/s, one after the other 19.7 k) but overlaps well with another
        # stream's launches (23.4 k): such trials go through the pool one by one, on the single-trial kernels
        n = int(cfgs[0]["N"])
        if n_workers > 1 and ((n + 15) // 16) ** 2 >= n_cus:
            parts = len(cfgs)
        # sizes the small-N trial kernel advances (SG: N <= 79; FSG: the coarse levels, where the time goes) run every
        # trial on an XCD of its own inside ONE launch: halves on two streams would only take turns at the chip
        sv = cfgs[0]["solver"]
        small = n if not sv["_target_"].endswith("FSGSolver") else (n // 2 if int(sv.get("n_levels", 1)) > 1 and n // 2 >= 12 else n)
        if ((small + 16) // 16) ** 2 <= 25 and int(sv.get("persistent", -1)) in (-1, 3):
            parts = 1
        cut = [(len(cfgs) * k) // parts for k in range(parts + 1)]
        for k in range(parts):
            weight = max(trial_cost(dict(N=c["N"], Re=c["Re"]), solver=c["solver"].get("_target_", "")) for c in cfgs[cut[k]:cut[k + 1]])
            tasks.append((weight, gi, cut[k], cut[k + 1]))      # a batch lasts as long as its longest trial
    tasks.sort(key=lambda t: (-t[0], t[1], t[2]))
    n_tasks = len(tasks)
    n_threads = max(1, min(n_workers, n_tasks))
    lock, done, early = threading.Lock(), {}, {}
    t0 = time.perf_counter()

    def worker(_):
        while True:
            with lock:
                if not tasks:
                    return
                _, gi, lo, hi = tasks.pop(0)
            part = groups[gi][0][lo:hi]
            try:
                nodes = []
                for cfg in part:
                    node = {k: v for k, v in cfg["solver"].items() if k != "_target_"}
                    if device is not None:
                        node["device"] = device
                    # (launches that need co-resident work-groups -- the small-N trial kernel and the other persistent
                    #  modes -- are kept from overlapping each other by ldc_lib.resident_lock inside the solvers; the
                    #  launch-path batches of the other worker streams still run beside them)
                    nodes.append(node)
                fsg = part[0]["solver"]["_target_"].endswith("FSGSolver")
                if len(nodes) == 1:                 # alone on its stream: the single-trial kernels (no argument blocks in memory)
                    batch = _OneTrial(C.instantiate(dict(nodes[0], _target_=part[0]["solver"]["_target_"])))
                else:
                    batch = (BatchedFSGSolver if fsg else BatchedSGSolver)(nodes)     # built on this worker's stream
                batch.solve()
                done[(gi, lo)] = batch
                # The records of this batch (validation, Ghia error, results.json, solution.vts: host work on the trials'
                # own fields, ~45 ms each) right away, on this worker: the other workers' batches keep the GPU busy
                # meanwhile.  (300 records made after the last batch of a round were 13 s of a 34-s round.)  With MLflow
                # active they are made afterwards, in the caller's thread and in order (its fluent API keeps ONE active run).
                if _mlflow() is None:
                    cfgs_g, dirs_g = groups[gi]
                    try:
                        early[(gi, lo)] = [make_record(cfgs_g[lo + q], s, dirs_g[lo + q], t0) for q, s in enumerate(batch.solvers)]
                    except Exception:                # the solve stands; its records are tried again, one by one, afterwards
                        log.exception("records of a batch of %d trials at N=%s failed on the worker", len(part), part[0]["N"])
            except Exception as exc:                # the other batches go on; the farm reports the failure
                log.exception("batch of %d trials at N=%s failed", len(part), part[0]["N"])
                done[(gi, lo)] = exc

    wall = run_concurrently(list(range(n_threads)), worker, device)
    good = [b for b in done.values() if not isinstance(b, Exception)]
    busy = sum(b.batch_seconds for b in good)
    out = []
    for gi, (cfgs, dirs) in enumerate(groups):
        recs = []
        for (g2, lo), batch in sorted(((k, v) for k, v in done.items() if k[0] == gi), key=lambda kv: kv[0][1]):
            if isinstance(batch, Exception):
                hi = min([k[1] for k in done if k[0] == gi and k[1] > lo] + [len(cfgs)])
                recs += [dict(error=repr(batch), objective=math.inf) for _ in range(lo, hi)]
                continue
            for q, s in enumerate(batch.solvers):
                scale = wall / busy if (n_threads > 1 and busy > 0) else 1.0      # the shares of all trials add up to the pool's wall time
                s.metrics.wall_time_seconds *= scale
                if (gi, lo) in early:               # made on the worker: bring the wall-time share up to date
                    r = early[(gi, lo)][q]
                    r["metrics"] = s.metrics.to_mlflow()
                    m = s.metrics
                    r["steps_per_second"] = m.iterations / m.wall_time_seconds if m.wall_time_seconds > 0 else 0.0
                else:
                    r = make_record(cfgs[lo + q], s, dirs[lo + q], t0)
                # the batch's own wall time (batches of other streams ran beside it) and the pool's
                r["solve_batch_seconds"], r["solve_batch_size"] = batch.batch_seconds, len(batch)
                r["solve_pool_seconds"], r["solve_streams"], r["solve_group_size"] = wall, n_threads, len(cfgs)
                if (gi, lo) in early:
                    (dirs[lo + q] / "results.json").write_text(json.dumps(_jsonable(r), indent=1))
                recs.append(r)
            batch.close()
        out.append(recs)
    return out


class _OneTrial:
    """A "batch" of one: the plain solver behind the interface run_batches drives."""

    def __init__(self, solver):
        self.solvers = [solver]
        self.batch_seconds, self.batch_size = 0.0, 1

    def __len__(self):
        return 1

    def solve(self):
        t0 = time.perf_counter()
        self.solvers[0].solve()
        self.batch_seconds = time.perf_counter() - t0

    def close(self):
        if hasattr(self.solvers[0], "close"):
            self.solvers[0].close()


_TRACKER = None        # utilities.tracking.sweep.SweepTracker of the running sweep (None: single run / no mlflow)


def _mlflow():
    try:
        import mlflow
        return mlflow
    except ImportError:
        return None


def _parent_of(cfg):
    from utilities.tracking import sweep as T
    if _TRACKER is not None:
        return _TRACKER.parents.get(T.resolve_sweep_name(_TRACKER.base_sweep_name, cfg))
    return None


def _ts_batch(solver):
    ts = getattr(solver, "time_series", None)
    if ts is not None and hasattr(ts, "to_mlflow_batch"):
        try:
            return ts.to_mlflow_batch()
        except Exception:                         # needs mlflow.entities; a stub without it is fine
            return None
    return None


def _mlflow_log(cfg, rec, solver):
    """Tracking of one finished trial (reference main.py:75-119), active only when mlflow is installed.  When a
    run is already active (run_solver opened it before solve(), like the reference) the results go into it;
    otherwise (batched trials) a child run is written after the fact.  In a sweep the run is nested under the
    parent run of its ``sweep_name`` (utilities.tracking.sweep)."""
    mlflow = _mlflow()
    if mlflow is None:
        return
    from utilities.tracking import sweep as T
    active = mlflow.active_run()
    if active is not None:
        T.log_results(mlflow, rec, active.info.run_id, _ts_batch(solver))
        return
    if _TRACKER is None:
        mlflow.set_tracking_uri(cfg.get("mlflow", {}).get("tracking_uri", "./mlruns"))
        mlflow.set_experiment(T.experiment_name(cfg))
    T.log_child_run(mlflow, cfg, rec, parent_id=_parent_of(cfg), time_series_batch=_ts_batch(solver))


def hydra_main(argv: list) -> float | None:
    """The reference's own launch path, for machines where Hydra (and its joblib / Optuna plugins, MLflow) are
    installed: ``@hydra.main`` composes ``conf/`` exactly as the reference's ``main.py:228-252`` does, Hydra's
    launcher / sweeper run one job per trial, ``utilities.mlflow.callback.MLflowSweepCallback`` keeps the parent
    runs, and each job is ``run_solver`` below -- same solver plugin, same records.  (Trials do not share launches
    on this path: Hydra starts one process per job.  The built-in launcher above is what batches them.)"""
    import hydra
    from hydra.core.hydra_config import HydraConfig
    from omegaconf import OmegaConf

    @hydra.main(config_path=str(HERE / "conf"), config_name="config", version_base=None)
    def _job(cfg):
        plain = OmegaConf.to_container(cfg, resolve=True)
        try:
            out_dir = Path(HydraConfig.get().runtime.output_dir)
        except Exception:                                   # outside a Hydra job (tests with a stand-in)
            out_dir = Path("hydra_outputs/run")
        rec = run_solver(plain, out_dir)
        return rec["objective"]                             # the Optuna sweeper minimises the job's return value

    saved = sys.argv
    sys.argv = [saved[0]] + list(argv)
    try:
        return _job()
    finally:
        sys.argv = saved


def main(argv=None) -> float | None:
    argv = list(sys.argv[1:] if argv is None else argv)
    logging.basicConfig(level=logging.INFO, format="[%(asctime)s][%(name)s] %(message)s")
    if "--hydra" in argv or os.environ.get("LDC_LAUNCHER", "").lower() == "hydra":
        return hydra_main([a for a in argv if a != "--hydra"])
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
    device, shared_card = None, False
    if dist.world > 1:          # one GPU per rank; ranks beyond the visible cards share them (tests: 2 ranks, 1 card)
        import torch
        n_cards = max(1, torch.cuda.device_count())
        device = f"cuda:{dist.local_rank % n_cards}"
        shared_card = int(os.environ.get("LOCAL_WORLD_SIZE", dist.world)) > n_cards
        if shared_card:
            log.warning("%d ranks on %d visible GPU(s): ranks share cards, every trial takes the launch path "
                        "(solver.persistent=0)", dist.world, n_cards)

    stamp_cfg = C.resolve(C.compose_job(composer, overrides, fixed))
    hy = stamp_cfg.get("hydra", {})
    root_tpl = (hy.get("sweep", {}) if multirun else hy.get("run", {})).get("dir", "hydra_outputs/run")
    root_dir = time.strftime(str(root_tpl).replace("${now:", "").replace("}", ""), time.localtime())
    root_dir = Path(dist.all_gather_object(root_dir)[0])          # every rank uses rank 0's timestamp

    SG, FSG = "solvers.spectral.sg.SGSolver", "solvers.spectral.fsg.FSGSolver"
    batch_cap_given = "LDC_MAX_BATCH" in os.environ or "batch_trials" in (stamp_cfg.get("hydra", {}).get("launcher", {}) or {})
    max_batch = int(os.environ.get("LDC_MAX_BATCH", stamp_cfg.get("hydra", {}).get("launcher", {}).get("batch_trials", 64)))

    # Which kernel advances a trial depends, in auto mode, on the size AND on how many trials share its batch (batched.py);
    # the kernels agree to rounding, so an iteration count at the stopping threshold can move by one with LDC_MAX_BATCH, the
    # world size or the search-round size.  A study that must not depend on those pins ONE mode for every trial whose
    # configuration leaves the choice open (solver.persistent = -1): LDC_PIN_MODE=0 (launch path) or 3 (one XCD per trial
    # where the size fits, the launch path above); search_mode=reference pins 3.  Ranks that share a card (more ranks than
    # visible GPUs) get 0: co-resident launches of two PROCESSES cannot be kept apart by ldc_lib.resident_lock.
    sw0 = stamp_cfg.get("hydra", {}).get("sweeper", {}) or {}
    pin_mode = os.environ.get("LDC_PIN_MODE")
    if pin_mode is None and search and str(os.environ.get("LDC_SEARCH_MODE", sw0.get("search_mode", "throughput"))).lower() == "reference":
        pin_mode = 3
    if shared_card:
        pin_mode = 0
    if pin_mode is not None and int(pin_mode) not in (0, 3):
        raise ValueError(f"LDC_PIN_MODE={pin_mode}: 0 (launch path) or 3 (one XCD per trial where it fits)")

    def job_cfg(assignment, index):
        cfg = C.compose_job(composer, overrides, list(fixed) + list(assignment))
        cfg.setdefault("hydra", {}).setdefault("job", {})["num"] = index
        cfg = C.resolve(cfg)
        sv = cfg.get("solver") or {}
        if pin_mode is not None and str(sv.get("_target_", "")) in (SG, FSG) and (int(sv.get("persistent", -1)) == -1 or shared_card):
            sv["persistent"] = int(pin_mode)
        return cfg

    def run_group(items, jobs, offset=0):
        """items: [(index, trial)] owned by this rank with one group key.  Trials that can share their launches
        -- same solver class, N, level hierarchy and diagnostics flag -- are advanced together; each keeps its
        own Re / lid / tolerance / max_iterations (solvers.spectral.batched)."""
        cfgs = [job_cfg(jobs[i], offset + i) for i, _ in items]
        out = [None] * len(cfgs)
        share = {}
        for q, c in enumerate(cfgs):
            sv = c["solver"]
            share.setdefault((sv["_target_"], int(c["N"]), int(sv.get("n_levels", 0)), bool(sv.get("diagnostics", True)),
                              int(sv.get("nx", c["N"])), int(sv.get("ny", c["N"]))), []).append(q)      # (nx, ny: solver.ny=... overrides)
        groups, where = [], []
        for (target, _, _, _, _, _), members in share.items():
            if target in (SG, FSG) and len(members) > 1 and max_batch > 1:
                # sizes the trial-per-CU kernel holds (M <= 44): one work-group per trial, 256 advance at once -- a larger
                # batch is a better batch there (unless LDC_MAX_BATCH / hydra.launcher.batch_trials says otherwise)
                n_of = int(cfgs[members[0]]["N"])
                cap = max_batch if (batch_cap_given or n_of + 1 > 44) else max(max_batch, 256)
                for lo in range(0, len(members), cap):
                    part = members[lo: lo + cap]
                    log.info("batch of %d trials at N=%s on %s", len(part), cfgs[part[0]]["N"], device or "cuda:0")
                    groups.append(([cfgs[q] for q in part], [root_dir / str(offset + items[q][0]) for q in part]))
                    where.append(part)
            else:
                for q in members:
                    # one failing trial (an FV solver error, an LdcError, ...) costs its own record only: the other
                    # trials of this rank, finished or still to run, keep theirs (the farm's contract)
                    try:
                        out[q] = run_solver(cfgs[q], root_dir / str(offset + items[q][0]), device=device)
                    except Exception as exc:
                        log.exception("trial %d failed", offset + items[q][0])
                        out[q] = dict(error=repr(exc), objective=math.inf)
        if groups:          # all batches of this rank through ONE pool of streams: sizes overlap too
            for part, recs in zip(where, run_batches(groups, device)):
                for q, r in zip(part, recs):
                    out[q] = r
        for (i, _), r in zip(items, out):
            r["overrides"] = {k: v for k, v in jobs[i]}
        return out

    def key_of(jobs):
        # ONE group per rank: run_group sorts the rank's trials into batches by itself (solver class, N, hierarchy,
        # diagnostics) and advances all of them through one pool of streams
        return lambda t: 0

    def open_parents(jobs, offset=0):
        """Job start of the reference's MLflowSweepCallback for every job of this round: rank 0 gets or creates
        the parent run of each job's sweep_name, the other ranks adopt the map."""
        if _TRACKER is None:
            return
        if dist.rank == 0:
            for i, a in enumerate(jobs):
                _TRACKER.parent_for(job_cfg(a, offset + i))
        _TRACKER.adopt(dist.all_gather_object(_TRACKER.parents if dist.rank == 0 else {})[0])

    solver_hint = str((stamp_cfg.get("solver") or {}).get("_target_", ""))

    def cost_of(trial, batch=1):     # expected GPU seconds (measured iteration counts x time per iteration, by solver class and batch size)
        return trial_cost(dict({"Re": stamp_cfg.get("Re", 100)}, **trial), solver=solver_hint, batch=batch)

    global _TRACKER
    _TRACKER = None
    if multirun:
        from utilities.tracking.sweep import SweepTracker
        _TRACKER = SweepTracker.create(stamp_cfg)
        if _TRACKER is not None:
            _TRACKER.start(stamp_cfg, raw_sweep_name=base_cfg.get("sweep_name"))

    if not search:
        jobs = C.expand_grid(space)
        open_parents(jobs)
        trials = [dict(a, N=dict(a).get("N", base_cfg.get("N", 32)), _job=i) for i, a in enumerate(jobs)]
        failure = None
        try:
            recs = run_farm(trials, None, dist, run_group=lambda items: run_group(items, jobs), group_key=key_of(jobs),
                            cost=cost_of)
        except FarmError as exc:        # finished trials are kept and written out below, then the error surfaces
            recs, failure = exc.records, exc
        objective = recs[0]["objective"] if len(recs) == 1 else None
    else:
        sw = stamp_cfg.get("hydra", {}).get("sweeper", {}) or {}
        n_trials = int(sw.get("n_trials", 15))
        # Candidates per round (utilities.sweep.farm.plan_rounds): a sampler learns only between rounds, so the round
        # size is part of the search.  mode "reference": rounds of the experiment's n_jobs whatever the world size (the
        # reference's ask n_jobs / run / tell n_jobs); mode "throughput" (default): every GPU is offered
        # `trials_per_gpu` candidates (default n_jobs) per round -- ONE GPU advances a batch of equal-N trials with the
        # same launches at little more than the cost of one -- but a study never has fewer than three rounds, so
        # that model-guided rounds remain (one round of n_trials candidates is a random search).  With one GPU both
        # modes give the reference's sequence.  The sampler's seed is fixed (the reference's is unset).
        per_gpu = int(os.environ.get("LDC_TRIALS_PER_GPU", sw.get("trials_per_gpu", sw.get("n_jobs", 1))))
        mode = str(os.environ.get("LDC_SEARCH_MODE", sw.get("search_mode", "throughput"))).lower()
        rounds = plan_rounds(n_trials, int(sw.get("n_jobs", 1)), dist.world, per_gpu=per_gpu, mode=mode,
                             min_rounds=int(os.environ.get("LDC_MIN_ROUNDS", sw.get("min_rounds", 3))))
        if len(rounds) < 3:
            log.warning("this study runs in %d round(s) of %s candidates: the sampler is told results only between "
                        "rounds, so fewer than 3 rounds is (nearly) a random search", len(rounds), rounds)
        log.info("search: %d trials in %d rounds %s (mode %s, %d GPU(s), %d candidates per GPU and round at most)",
                 n_trials, len(rounds), rounds, mode, dist.world, per_gpu)
        seed = int((sw.get("sampler") or {}).get("seed", 0))
        sampler = TPESampler(space, seed=seed)
        recs, done, failure = [], 0, None
        for rnd, size in enumerate(rounds):
            guided = sampler.is_guided()                  # does this round come from the model or from the prior?
            batch = [sampler.ask() for _ in range(size)]
            jobs = [list(b.items()) for b in batch]
            open_parents(jobs, done)
            trials = [dict(b, N=b.get("N", base_cfg.get("N", 32)), _job=i) for i, b in enumerate(batch)]
            # a failed trial is a failed trial (objective inf), like an exception inside an Optuna objective
            out = run_farm(trials, None, dist, run_group=lambda items: run_group(items, jobs, done),
                           group_key=key_of(jobs), raise_on_error=False, cost=cost_of)
            for b, r in zip(batch, out):
                sampler.tell(b, r["objective"] if isinstance(r["objective"], (int, float)) else math.inf)
                r.update(search_round=rnd, search_round_size=size, search_rounds=len(rounds), search_mode=mode,
                         search_model_guided=bool(guided))
            recs.extend(out)
            done += len(batch)
        best, val = sampler.best
        log.info("best trial: %s -> %s", best, val)
        objective = val
    if _TRACKER is not None:
        if dist.rank == 0:
            _TRACKER.finish(stamp_cfg, recs, is_search=bool(search))
        _TRACKER = None
    if dist.rank == 0 and (len(recs) > 1 or search):
        root_dir.mkdir(parents=True, exist_ok=True)
        (root_dir / "sweep_results.json").write_text(json.dumps(_jsonable(recs), indent=1))
        log.info("gathered %d trial records -> %s", len(recs), root_dir / "sweep_results.json")
    dist.barrier()
    dist.close()
    if failure is not None:
        raise failure
    return objective


if __name__ == "__main__":
    main()
