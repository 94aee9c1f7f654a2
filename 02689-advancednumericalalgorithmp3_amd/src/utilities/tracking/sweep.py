"""Parent-run bookkeeping of a sweep: one MLflow parent run per ``sweep_name``, children nested under it.

Contract: reference ``src/utilities/mlflow/callback.py`` (``MLflowSweepCallback``: ``on_multirun_start``
:135-188, ``_get_or_create_parent`` :89-133, ``on_job_start`` :190-217, ``_log_optuna_results_to_parent``
:219-314, ``on_multirun_end`` :316-327) and the child-run side in ``main.py:75-96``.  The reference hangs this
on Hydra's callback hooks; the launcher here (``main.py``) has no Hydra underneath, so the same three moments
are plain method calls:

    tracker = SweepTracker.create(cfg)      # None when mlflow is not installed
    tracker.start(cfg)                      # multirun start: tracking URI, experiment, MLFLOW_SWEEP_ACTIVE
    pid = tracker.parent_for(job_cfg)       # job start: get-or-create the parent of this job's sweep_name,
                                            #            exports MLFLOW_PARENT_RUN_ID for the child run
    tracker.finish(cfg, records)            # multirun end: trial table + best trial on the parent, env cleaned

With several ranks (one per GPU) only rank 0 talks to the tracking server for the parents; the launcher
gathers the ``{sweep_name: run_id}`` map and the other ranks ``adopt`` it, so that every child run, whichever
GPU it ran on, is nested under the same parent.
"""
from __future__ import annotations

import logging
import os
import re

log = logging.getLogger(__name__)


def experiment_name(cfg: dict) -> str:
    """``project_prefix/experiment_name`` unless the name is absolute (reference main.py:28-34)."""
    name = str(cfg.get("experiment_name", "LDC-Dev"))
    prefix = (cfg.get("mlflow") or {}).get("project_prefix", "")
    return f"{prefix}/{name}" if prefix and not name.startswith("/") else name


def resolve_sweep_name(base: str, cfg: dict) -> str:
    """``my-sweep-Re${Re}`` -> one parent per Reynolds number (callback.py:203-210)."""
    name = str(base or cfg.get("sweep_name", "sweep"))
    if "${Re}" in name or "{Re}" in name:
        re_value = str(int(cfg.get("Re", 100)))
        name = name.replace("${Re}", re_value).replace("{Re}", re_value)
    return name


class SweepTracker:
    def __init__(self, mlflow_module):
        self.mlflow = mlflow_module
        self.parents: dict = {}           # sweep_name -> run_id
        self.tracking_uri = None
        self.experiment = None
        self.base_sweep_name = None
        self.active = False

    @classmethod
    def create(cls, cfg: dict = None):
        try:
            import mlflow
        except ImportError:
            return None
        return cls(mlflow)

    # ------------------------------------------------------------------ multirun start
    def start(self, cfg: dict, raw_sweep_name: str = None):
        ml = self.mlflow
        mcfg = cfg.get("mlflow") or {}
        self.tracking_uri = mcfg.get("tracking_uri", "./mlruns")
        if str(mcfg.get("mode", "")).lower() in ("files", "local"):
            os.environ.pop("MLFLOW_TRACKING_URI", None)
        os.environ["MLFLOW_TRACKING_URI"] = str(self.tracking_uri)
        ml.set_tracking_uri(self.tracking_uri)
        self.experiment = experiment_name(cfg)
        try:
            ml.set_experiment(self.experiment)
        except Exception as exc:                          # callback.py:160-170
            fallback = f"{self.experiment}-restored"
            log.warning("MLflow set_experiment failed for '%s' (%s); falling back to '%s'",
                        self.experiment, exc, fallback)
            self.experiment = fallback
            ml.set_experiment(self.experiment)
        self.base_sweep_name = raw_sweep_name if raw_sweep_name is not None else cfg.get("sweep_name", "sweep")
        os.environ["MLFLOW_SWEEP_ACTIVE"] = "1"
        self.active = True
        log.info("MLflow sweep tracking initialised for experiment: %s", self.experiment)

    # ------------------------------------------------------------------ job start
    def _find_existing_parent(self, sweep_name: str):
        try:
            runs = self.mlflow.search_runs(
                experiment_names=[self.experiment],
                filter_string=f"tags.sweep = 'parent' AND tags.`mlflow.runName` = '{sweep_name}'",
                order_by=["start_time DESC"], max_results=1)
            if getattr(runs, "empty", True):
                return None
            return runs.iloc[0]["run_id"]
        except Exception as exc:
            log.warning("Error searching for parent run: %s", exc)
            return None

    def parent_for(self, job_cfg: dict) -> str | None:
        """Parent run id of this job's sweep (created on first use); exported as MLFLOW_PARENT_RUN_ID."""
        if not self.active:
            return None
        ml = self.mlflow
        name = resolve_sweep_name(self.base_sweep_name, job_cfg)
        pid = self.parents.get(name)
        if pid is None:
            pid = self._find_existing_parent(name)
            if pid:
                log.info("Reusing existing parent run '%s': %s", name, pid)
            else:
                run = ml.start_run(run_name=name)
                pid = run.info.run_id
                try:
                    ml.log_dict({k: v for k, v in job_cfg.items() if k != "hydra"}, "sweep_config.yaml")
                except Exception as exc:                  # a config that does not serialise must not stop a sweep
                    log.warning("could not log the sweep config: %s", exc)
                ml.set_tag("sweep", "parent")
                m = re.search(r"Re(\d+)", name)
                if m:
                    ml.set_tag("Re", m.group(1))
                job_id = os.environ.get("LSB_JOBID")
                if job_id:
                    ml.set_tag("lsf.job_id", job_id)
                    ml.set_tag("lsf.job_name", os.environ.get("LSB_JOBNAME", ""))
                ml.end_run()                              # referenced by id from now on
                log.info("Created parent run '%s': %s", name, pid)
            self.parents[name] = pid
        os.environ["MLFLOW_PARENT_RUN_ID"] = pid
        return pid

    def adopt(self, parents: dict):
        """Take over the ``{sweep_name: run_id}`` map another rank created."""
        self.parents.update(parents or {})

    # ------------------------------------------------------------------ multirun end
    def finish(self, cfg: dict, records: list, is_search: bool = False):
        if not self.active:
            return
        os.environ.pop("MLFLOW_PARENT_RUN_ID", None)
        os.environ.pop("MLFLOW_SWEEP_ACTIVE", None)
        self.active = False
        log.info("Multirun sweep completed")
        if is_search and self.parents and records:
            self._log_search_results(next(iter(self.parents.values())), records)

    def _log_search_results(self, parent_id: str, records: list):
        """Trial table and best trial on the parent run (callback.py:219-314; the table is built from the
        gathered trial records instead of a search over the child runs -- same columns)."""
        ml = self.mlflow
        rows = []
        for r in records:
            if "error" in r:
                continue
            ve, m, p = r.get("validation_errors") or {}, r.get("metrics") or {}, r.get("params") or {}
            u, v = ve.get("u_L2_error"), ve.get("v_L2_error")
            inf = float("inf")
            rows.append(dict(
                trial=r.get("trial_index"), corner_smoothing=p.get("corner_smoothing"), u_L2_error=u, v_L2_error=v,
                iterations=m.get("iterations"), converged=m.get("converged"), wall_time=m.get("wall_time_seconds"),
                objective=r.get("objective"),
                combined_L2=((inf if u is None else u) ** 2 + (inf if v is None else v) ** 2) ** 0.5))
        if not rows:
            log.warning("No completed trials for the search summary")
            return
        best = min(rows, key=lambda q: q["combined_L2"])
        try:
            with ml.start_run(run_id=parent_id, nested=False):
                try:
                    import pandas as pd
                    ml.log_table(pd.DataFrame(rows), artifact_file="optuna_trials.json")
                except ImportError:
                    ml.log_dict({"trials": rows}, "optuna_trials.json")
                ml.log_metrics({
                    "best_corner_smoothing": float(best.get("corner_smoothing") or 0.0),
                    "best_u_L2_error": float(best["u_L2_error"] if best["u_L2_error"] is not None else float("inf")),
                    "best_v_L2_error": float(best["v_L2_error"] if best["v_L2_error"] is not None else float("inf")),
                    "best_combined_L2": float(best["combined_L2"])})
                ml.set_tag("best_trial", str(best.get("trial")))
                ml.log_metric("n_trials_completed", len(rows))
                ml.log_metric("n_trials_converged", int(sum(1 for q in rows if q.get("converged"))))
            log.info("Logged search results to parent run %s", str(parent_id)[:8])
        except Exception as exc:
            log.warning("Failed to log search results: %s", exc)


def child_tags(solver_name: str, parent_id: str = None) -> dict:
    parent = parent_id or os.environ.get("MLFLOW_PARENT_RUN_ID")
    tags = {"solver": solver_name}
    if parent:
        tags.update({"mlflow.parentRunId": parent, "parent_run_id": parent, "sweep": "child"})
    return tags


def open_child_run(mlflow, solver_name: str, run_name: str, parent_id: str = None):
    """Start the child run BEFORE solve(), as the reference does (main.py:92): ``mlflow.active_run()`` is then
    true inside the iteration loop and the live metrics of base.py:294-309 have a run to go to."""
    tags = child_tags(solver_name, parent_id)
    return mlflow.start_run(run_name=run_name, tags=tags, nested="parent_run_id" in tags)


def log_results(mlflow, rec: dict, run_id: str = None, time_series_batch=None):
    """Parameters, validation errors, final metrics and the down-sampled histories of one finished trial into the
    active run (reference main.py:93-110)."""
    mlflow.log_params(rec["params"])
    if rec.get("validation_errors"):
        mlflow.log_metrics(rec["validation_errors"])
    mlflow.log_metrics({k: v for k, v in rec["metrics"].items() if isinstance(v, (int, float))})
    if time_series_batch and run_id:
        try:
            mlflow.tracking.MlflowClient().log_batch(run_id, metrics=time_series_batch)
        except Exception as exc:
            log.warning("time-series batch not logged: %s", exc)


def log_child_run(mlflow, cfg: dict, rec: dict, parent_id: str = None, time_series_batch=None):
    """A whole child run after the fact (batched trials: they share their launches, so there is no single
    active run while they iterate)."""
    with open_child_run(mlflow, rec["solver"], rec["run_name"], parent_id) as run:
        log_results(mlflow, rec, run.info.run_id, time_series_batch)
        return run.info.run_id
