"""MLflow glue of the launcher against a stub ``mlflow`` module (the package is not installed here):
parent run per sweep_name (reference src/utilities/mlflow/callback.py:89-217), nested child runs
(main.py:75-96), the trial table on the parent (callback.py:219-314) and the live metrics of the iteration
loop (base.py:288-309).  CPU only."""
import os
import sys
import types

import numpy as np
import pytest


class _Runs:
    def __init__(self, rows):
        self._rows = rows
        self.empty = not rows

    @property
    def iloc(self):
        return self._rows


class FakeMlflow(types.ModuleType):
    """Just enough of the mlflow API surface the launcher touches; records every call."""

    def __init__(self):
        super().__init__("mlflow")
        self.runs = {}            # run_id -> dict(name, tags, params, metrics, tables, dicts, batches)
        self.stack = []
        self.calls = []
        self.uri = self.experiment = None
        fake = self

        class _Client:
            def log_batch(self, run_id, metrics=()):
                fake.runs[run_id]["batches"].append(list(metrics))

        self.tracking = types.SimpleNamespace(MlflowClient=_Client)
        ent = types.ModuleType("mlflow.entities")
        ent.Metric = lambda key, value, timestamp, step: types.SimpleNamespace(key=key, value=value, step=step)
        self.entities = ent

    # -- configuration
    def set_tracking_uri(self, uri):
        self.uri = uri

    def set_experiment(self, name):
        self.experiment = name

    # -- runs
    def start_run(self, run_name=None, tags=None, nested=False, run_id=None):
        if run_id is None:
            run_id = f"run{len(self.runs):04d}"
            self.runs[run_id] = dict(name=run_name, tags=dict(tags or {}), params={}, metrics={}, tables={},
                                     dicts={}, batches=[], nested=nested)
            self.runs[run_id]["tags"]["mlflow.runName"] = run_name
        self.stack.append(run_id)
        fake = self

        class _Ctx:
            info = types.SimpleNamespace(run_id=run_id)

            def __enter__(self):
                return self

            def __exit__(self, *a):
                fake.end_run()
                return False

        return _Ctx()

    def end_run(self):
        if self.stack:
            self.stack.pop()

    def active_run(self):
        if not self.stack:
            return None
        return types.SimpleNamespace(info=types.SimpleNamespace(run_id=self.stack[-1]))

    def _cur(self):
        return self.runs[self.stack[-1]]

    def set_tag(self, k, v):
        self._cur()["tags"][k] = v

    def log_params(self, p):
        self._cur()["params"].update(p)

    def log_metrics(self, m, step=None):
        self._cur()["metrics"].update(m)

    def log_metric(self, k, v):
        self._cur()["metrics"][k] = v

    def log_dict(self, d, name):
        self._cur()["dicts"][name] = d

    def log_table(self, df, artifact_file):
        self._cur()["tables"][artifact_file] = df

    def search_runs(self, experiment_names=None, filter_string="", order_by=None, max_results=None):
        import re
        m = re.search(r"`mlflow.runName` = '([^']*)'", filter_string)
        rows = [dict(run_id=rid) for rid, r in self.runs.items()
                if r["tags"].get("sweep") == "parent" and (m is None or r["name"] == m.group(1))]
        return _Runs(rows)


@pytest.fixture()
def fake_mlflow(monkeypatch):
    fm = FakeMlflow()
    monkeypatch.setitem(sys.modules, "mlflow", fm)
    monkeypatch.setitem(sys.modules, "mlflow.entities", fm.entities)
    for k in ("MLFLOW_PARENT_RUN_ID", "MLFLOW_SWEEP_ACTIVE", "MLFLOW_TRACKING_URI"):
        monkeypatch.delenv(k, raising=False)
    return fm


def _cfg(**kw):
    base = dict(experiment_name="LDC-Dev", sweep_name="dev-run", Re=100, N=32,
                mlflow=dict(mode="files", project_prefix="", tracking_uri="./mlruns"))
    base.update(kw)
    return base


def test_parent_run_is_created_once_per_sweep_name_and_exported(fake_mlflow):
    from utilities.tracking.sweep import SweepTracker, experiment_name
    assert experiment_name(_cfg(mlflow=dict(project_prefix="/Shared"))) == "/Shared/LDC-Dev"
    assert experiment_name(_cfg(experiment_name="/abs", mlflow=dict(project_prefix="/Shared"))) == "/abs"
    tr = SweepTracker.create(_cfg())
    assert tr is not None
    tr.start(_cfg(), raw_sweep_name="sweep-Re${Re}")
    assert os.environ["MLFLOW_SWEEP_ACTIVE"] == "1" and os.environ["MLFLOW_TRACKING_URI"] == "./mlruns"
    assert fake_mlflow.experiment == "LDC-Dev"
    p100 = tr.parent_for(_cfg(Re=100))
    assert os.environ["MLFLOW_PARENT_RUN_ID"] == p100
    assert tr.parent_for(_cfg(Re=100, N=64)) == p100                 # reused from the cache
    p400 = tr.parent_for(_cfg(Re=400))
    assert p400 != p100 and os.environ["MLFLOW_PARENT_RUN_ID"] == p400
    par = fake_mlflow.runs[p100]
    assert par["name"] == "sweep-Re100" and par["tags"]["sweep"] == "parent" and par["tags"]["Re"] == "100"
    assert "sweep_config.yaml" in par["dicts"] and fake_mlflow.active_run() is None      # parent closed again
    # a second launcher process finds the parent that already exists in the store instead of a new one
    tr2 = SweepTracker.create(_cfg())
    tr2.start(_cfg(), raw_sweep_name="sweep-Re${Re}")
    assert tr2.parent_for(_cfg(Re=400)) == p400
    n_before = len(fake_mlflow.runs)
    tr2.adopt({"other": "runXXXX"})
    assert tr2.parents["other"] == "runXXXX" and len(fake_mlflow.runs) == n_before
    tr.finish(_cfg(), [])
    assert "MLFLOW_PARENT_RUN_ID" not in os.environ and "MLFLOW_SWEEP_ACTIVE" not in os.environ


def test_child_runs_are_nested_and_the_search_summary_lands_on_the_parent(fake_mlflow):
    from utilities.tracking import sweep as T
    tr = T.SweepTracker.create(_cfg())
    tr.start(_cfg(sweep_name="corner-smoothing-fv_l2_error"))
    pid = tr.parent_for(_cfg())
    recs = []
    for k, (cs, u, v) in enumerate(((0.05, 0.3, 0.4), (0.15, 0.03, 0.04), (0.30, 0.1, 0.2))):
        rec = dict(run_name="spectral_fsg_N129", solver="spectral_fsg", trial_index=k,
                   params=dict(corner_smoothing=cs, nx=128), objective=(u * u + v * v) ** 0.5,
                   metrics=dict(iterations=100 + k, converged=True, wall_time_seconds=1.0),
                   validation_errors=dict(u_L2_error=u, v_L2_error=v))
        rid = T.log_child_run(fake_mlflow, _cfg(), rec, parent_id=pid)
        child = fake_mlflow.runs[rid]
        assert child["nested"] and child["tags"]["mlflow.parentRunId"] == pid and child["tags"]["sweep"] == "child"
        assert child["tags"]["parent_run_id"] == pid and child["tags"]["solver"] == "spectral_fsg"
        assert child["params"]["corner_smoothing"] == cs and child["metrics"]["u_L2_error"] == u
        recs.append(rec)
    recs.append(dict(error="RuntimeError('boom')", objective=float("inf"), trial_index=3))     # a failed trial
    tr.finish(_cfg(), recs, is_search=True)
    par = fake_mlflow.runs[pid]
    assert par["metrics"]["best_corner_smoothing"] == 0.15 and par["metrics"]["n_trials_completed"] == 3
    assert par["metrics"]["best_combined_L2"] == pytest.approx(0.05)
    assert par["metrics"]["n_trials_converged"] == 3 and "optuna_trials.json" in par["tables"]


def test_live_metrics_every_50_iterations_go_to_the_active_run(fake_mlflow):
    """base.py:288-309: iteration i is reported when i % 50 == 0 or it converged; energy / enstrophy only after
    the 10 warm-up iterations."""
    from solvers.base import LidDrivenCavitySolver, REL, EN, ZN

    class Dummy(LidDrivenCavitySolver):
        def __init__(self):
            self.params = types.SimpleNamespace(diagnostics=True)

        _begin = _advance = _finalize_fields = compute_vortex_metrics = None

    Dummy.__abstractmethods__ = frozenset()
    s = Dummy()
    recs = np.zeros((130, 8))
    recs[:, REL] = np.arange(130) * 1e-3
    recs[:, EN] = 7.0
    recs[:, ZN] = 9.0
    s._live_log(recs, 0, False)                       # no active run: nothing is sent anywhere
    assert all(not r["batches"] for r in fake_mlflow.runs.values())
    with fake_mlflow.start_run(run_name="child") as run:
        s._live_log(recs, 0, False)
        s._live_log(recs[:77], 130, True)             # second chunk: iterations 130..206, converged at 206
    got = fake_mlflow.runs[run.info.run_id]["batches"]
    steps1 = sorted({m.step for m in got[0]})
    assert steps1 == [0, 50, 100]
    by_step = {st: {m.key: m.value for m in got[0] if m.step == st} for st in steps1}
    assert set(by_step[0]) == {"rel_iter_residual", "u_residual", "v_residual", "continuity_residual"}
    assert by_step[50]["energy"] == 7.0 and by_step[50]["enstrophy"] == 9.0
    assert by_step[100]["rel_iter_residual"] == pytest.approx(0.1)
    assert sorted({m.step for m in got[1]}) == [150, 200, 206]


def test_abi_calls_run_on_the_solvers_own_device(monkeypatch):
    """SGSolver(device='cuda:1') must launch on GPU 1 whatever the caller's current device is: every C-ABI
    launch goes through _abi, which makes the solver's device current and hands over ITS current stream."""
    import torch
    from solvers.spectral import ldc_lib as L
    from solvers.spectral.sg import SGSolver
    seen = {}

    class Ctx:
        def __init__(self, dev):
            seen["ctx"] = dev

        def __enter__(self):
            seen["inside"] = True

        def __exit__(self, *a):
            seen["inside"] = False

    def fake_stream_ptr(device=None):
        seen["stream_for"] = device
        return 0xBEEF

    def fake_pack(*args):
        seen["args"], seen["was_inside"] = args, seen.get("inside")
        return 0

    monkeypatch.setattr(torch.cuda, "device", Ctx)
    monkeypatch.setattr(L, "stream_ptr", fake_stream_ptr)
    monkeypatch.setattr(L, "lib", lambda: types.SimpleNamespace(ldc_pack=fake_pack))
    s = SGSolver.__new__(SGSolver)
    s.device = torch.device("cuda:1")
    s._abi("ldc_pack", 1, 2, 3)
    assert seen["ctx"] == torch.device("cuda:1") and seen["stream_for"] == torch.device("cuda:1")
    assert seen["args"] == (1, 2, 3, 0xBEEF) and seen["was_inside"] is True
    # and no launch site bypasses the helper
    import inspect
    import solvers.spectral.sg as sg_mod
    import solvers.spectral.fsg as fsg_mod
    for mod in (sg_mod, fsg_mod):
        src = inspect.getsource(mod)
        assert "L.stream_ptr()" not in src


def test_real_hydra_path_is_wired_when_hydra_is_importable(fake_mlflow, monkeypatch, tmp_path):
    """``main.py --hydra`` hands the job to ``@hydra.main`` and the reference's callback target resolves
    (``utilities.mlflow.callback.MLflowSweepCallback``): checked with stand-in ``hydra`` / ``omegaconf`` modules
    (neither package is installed here) -- the decorator receives conf/ + "config", the job function turns the
    DictConfig into plain containers and runs ``run_solver``, its return value is the objective."""
    import importlib.util
    from conftest import PKG
    seen = {}

    class DictConfig(dict):
        pass

    omegaconf = types.ModuleType("omegaconf")
    omegaconf.OmegaConf = types.SimpleNamespace(to_container=lambda cfg, resolve=True: dict(cfg))
    omegaconf.DictConfig = DictConfig
    hydra = types.ModuleType("hydra")

    def hydra_main(config_path=None, config_name=None, version_base=None):
        seen.update(config_path=config_path, config_name=config_name)

        def deco(fn):
            def run():
                seen["argv"] = list(sys.argv[1:])
                return fn(DictConfig(N=16, Re=100, solver=dict(name="spectral"), sweep_name="s", experiment_name="e"))
            return run
        return deco

    hydra.main = hydra_main
    core = types.ModuleType("hydra.core")
    hc = types.ModuleType("hydra.core.hydra_config")
    hc.HydraConfig = types.SimpleNamespace(get=lambda: types.SimpleNamespace(runtime=types.SimpleNamespace(output_dir=str(tmp_path))))
    exp = types.ModuleType("hydra.experimental")
    cb = types.ModuleType("hydra.experimental.callback")
    cb.Callback = type("Callback", (), {})
    for name, mod in (("hydra", hydra), ("hydra.core", core), ("hydra.core.hydra_config", hc), ("omegaconf", omegaconf),
                      ("hydra.experimental", exp), ("hydra.experimental.callback", cb)):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.delitem(sys.modules, "utilities.mlflow.callback", raising=False)
    spec = importlib.util.spec_from_file_location("ldc_main_hydra", PKG / "main.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(mod, "run_solver", lambda cfg, out_dir, device=None: seen.update(cfg=cfg, out=out_dir) or {"objective": 0.25})
    assert mod.main(["--hydra", "solver=spectral", "N=16"]) == 0.25
    assert seen["config_name"] == "config" and seen["config_path"].endswith("conf")
    assert seen["argv"] == ["solver=spectral", "N=16"] and type(seen["cfg"]) is dict and seen["cfg"]["N"] == 16
    assert str(seen["out"]) == str(tmp_path)
    # the callback of the reference's config resolves and drives the same tracker
    from utilities.mlflow.callback import MLflowSweepCallback
    c = MLflowSweepCallback()
    cfg = DictConfig(experiment_name="E", sweep_name="hs", Re=100, mlflow=dict(mode="files", tracking_uri="./mlruns"))
    c.on_multirun_start(cfg)
    c.on_job_start(cfg)
    pid = os.environ["MLFLOW_PARENT_RUN_ID"]
    assert fake_mlflow.runs[pid]["name"] == "hs" and fake_mlflow.runs[pid]["tags"]["sweep"] == "parent"
    c.on_multirun_end(cfg)
    assert "MLFLOW_PARENT_RUN_ID" not in os.environ
