"""Host logic around the hot path: config composition (the reference's Hydra grammar), objectives,
Botella table, Ghia metric, VTS reader/writer.  CPU only."""
import json
import math
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import ldc_oracle as orc
from solvers import validation as V
from solvers.vtkio import StructuredGridFile, read_vts
from utilities.config import compose as C
from utilities.sweep.farm import TPESampler

from conftest import PKG


@pytest.fixture(scope="module")
def comp():
    return C.Composer(PKG / "conf")


def test_default_and_alias_resolve_to_sg(comp):
    for ov in ([], ["solver=spectral"], ["solver=spectral/sg"]):
        cfg = C.resolve(C.compose_job(comp, ov + ["N=256", "Re=1000"], []))
        s = cfg["solver"]
        assert s["_target_"] == "solvers.spectral.sg.SGSolver" and s["name"] == "spectral"
        assert (s["nx"], s["ny"], s["Re"]) == (256, 256, 1000)
        assert (s["CFL"], s["beta_squared"], s["corner_smoothing"], s["basis_type"]) == (1.5, 5.0, 0.15, "chebyshev")
        assert s["tolerance"] == 1e-6 and s["max_iterations"] == 10_000_000 and s["multigrid"] == "none"
    assert cfg["hydra"]["mode"] == "MULTIRUN" and cfg["validation"]["reference_dir"] == "data/validation/fv"


def test_constructor_keys_match_the_dataclass(comp):
    """Every key of conf/solver/spectral/*.yaml is a SpectralParameters field (TypeError otherwise)."""
    from dataclasses import fields
    from solvers.datastructures import SpectralParameters
    names = {f.name for f in fields(SpectralParameters)}
    for opt in ("spectral/sg", "spectral/fsg"):
        cfg = C.resolve(C.compose_job(comp, [f"solver={opt}"], []))
        assert set(cfg["solver"]) - {"_target_"} <= names


def test_experiment_overrides_and_sweep_space(comp):
    ov = ["+experiment/optimization=corner_smoothing", "solver.corner_smoothing=interval(0.02,0.35)",
          "optuna.objective=botella_vortex"]
    cfg, vals = comp.compose(ov)
    assert cfg["hydra"]["choices"]["solver"] == "spectral/fsg"
    assert cfg["hydra"]["sweeper"]["n_trials"] == 15 and cfg["hydra"]["sweeper"]["n_jobs"] == 5
    space, fixed = C.sweep_space(cfg, vals, True)
    assert space["N"] == [30, 40, 50] and space["solver.corner_smoothing"] == C.Interval(0.02, 0.35)
    assert fixed == [("optuna.objective", "botella_vortex")]
    job = C.resolve(C.compose_job(comp, ov, fixed + [("N", 40), ("solver.corner_smoothing", 0.07)]))
    assert job["solver"]["_target_"] == "solvers.spectral.fsg.FSGSolver"
    assert (job["solver"]["nx"], job["solver"]["corner_smoothing"], job["Re"]) == (40, 0.07, 1000)
    assert job["sweep_name"] == "corner-smoothing-botella_vortex" and job["max_iterations"] == 500000
    assert job["hydra"]["sweeper"]["study_name"].startswith("corner_smoothing_${optuna.objective}")  # hydra node stays raw


def test_grid_sweep_expansion_order(comp):
    cfg, vals = comp.compose(["N=64,128,256", "Re=100,400,1000"])
    space, _ = C.sweep_space(cfg, vals, True)
    grid = C.expand_grid(space)
    assert len(grid) == 9 and grid[0] == [("N", 64), ("Re", 100)] and grid[1] == [("N", 64), ("Re", 400)]
    cfg, vals = comp.compose(["+experiment/validation/saad-regu=spectral"])
    space, _ = C.sweep_space(cfg, vals, True)
    assert space == {"Re": [1000], "N": [16, 32, 64, 128]}
    job = C.resolve(C.compose_job(comp, ["+experiment/validation/saad-regu=spectral"], [("N", 64), ("Re", 1000)]))
    assert job["solver"]["corner_treatment"] == "saad" and job["validation"]["reference_dir"].endswith("fv-regu")
    cfg, vals = comp.compose(["+experiment/benchmarking=timings"])
    space, _ = C.sweep_space(cfg, vals, True)
    assert C.expand_grid(space) == [[("solver", "spectral/sg")], [("solver", "spectral/fsg")]]
    job = C.resolve(C.compose_job(comp, ["+experiment/benchmarking=timings"], [("solver", "spectral/fsg")]))
    assert job["solver"]["name"] == "spectral_fsg" and job["N"] == 15 and job["solver"]["n_levels"] == 2


def test_interpolation_and_errors(comp, monkeypatch):
    monkeypatch.setenv("MLFLOW_TRACKING_URI", "http://x")
    cfg = C.resolve(C.compose_job(comp, ["mlflow=coolify"], []))
    assert cfg["mlflow"]["tracking_uri"] == "http://x"
    with pytest.raises(C.ConfigError):
        comp.compose(["solver=does_not_exist"])
    assert C.parse_value("1e-6") == 1e-6 and C.parse_value("true") is True and C.parse_value("abc") == "abc"
    assert C.parse_sweep_value("30, 40, 50") == [30, 40, 50]
    assert C.parse_sweep_value("choice(a, b)") == ["a", "b"]


def test_objectives_follow_the_reference():
    assert V.compute_fv_l2_objective({"u_L2_error": 3.0, "v_L2_error": 4.0}) == 5.0
    assert V.compute_fv_l2_objective({"u_L2_error": 3.0}) == math.inf
    m = SimpleNamespace(psi_min=-0.1028946, psi_min_x=0.5975, psi_min_y=0.7357, omega_center=-3.1,
                        psi_BL=1e-6, omega_BL=0.0, psi_BL_x=0.03, psi_BL_y=0.03,
                        psi_BR=1.2e-5, omega_BR=0.0, psi_BR_x=0.94, psi_BR_y=0.06)
    ref = V.load_botella(100)
    want = math.sqrt((((m.psi_min - ref["psi_min"]) / ref["psi_min"]) ** 2 + (m.psi_min_x - ref["psi_min_x"]) ** 2
                      + (m.psi_min_y - ref["psi_min_y"]) ** 2) / 3)
    assert V.compute_botella_vortex_objective(m, 100) == pytest.approx(want, rel=1e-14)
    # quirk Q3: the Re=1000 table has other column names -> the reference objective is +inf
    assert V.compute_botella_vortex_objective(m, 1000, strict_reference_objective=True) == math.inf
    assert math.isfinite(V.compute_botella_vortex_objective(m, 1000))
    assert V.compute_botella_vortex_objective(m, 123) == math.inf
    with pytest.raises(ValueError, match="Multi-objective"):
        V.compute_optuna_objective("multi", {}, None, 100)
    rows = V.botella_table(m, 1000)
    assert len(rows) == 12 and rows[0]["Vortex"] == "Primary" and rows[0]["Botella"] == "0.118937"


def test_ghia_metric_matches_oracle_definition():
    o = orc.OracleSG(32, 100.0)
    for _ in range(400):
        o.step()
    (yu, ug), (xv, vg) = V.load_ghia(100)
    got = V.ghia_centerline_error(o.ax.x, o.ay.x, o.u, o.v, 100)
    want = orc.ghia_centerline_error(o.ax.x, o.ay.x, o.u, o.v, (yu, ug), (xv, vg))
    for k, v in want.items():
        assert got[k] == pytest.approx(v, rel=1e-12)
    assert len(yu) == 16 and ug[0] == 0.0 and ug[-1] == 1.0


def test_vts_roundtrip_and_reference_fixture(tmp_path):
    g = StructuredGridFile(np.linspace(0, 1, 5), np.linspace(0, 2, 4))
    g["u"] = np.arange(20.0)
    g["velocity"] = np.arange(60.0).reshape(20, 3)
    g.field_data["Re"] = np.array([400])
    g.save(tmp_path / "s.vts")
    r = read_vts(tmp_path / "s.vts")
    assert r["extent"] == (0, 4, 0, 3, 0, 0) and np.array_equal(r["point_data"]["u"], np.arange(20.0))
    assert r["point_data"]["velocity"].shape == (20, 3) and r["field_data"]["Re"][0] == 400
    assert np.allclose(r["points"][1], [0.25, 0.0, 0.0])
    # the packaged FV fixtures are the reference's solution.vts re-encoded (tools/convert_fv_reference.py)
    f = np.load(V.DATA_DIR / "fv" / "Re100" / "solution.npz")
    assert f["u"].size == 128 * 128 and abs(f["x"][0] - 0.5 / 128) < 1e-12 and f["u"].max() == pytest.approx(0.97617973912508)


def test_tpe_sampler_finds_a_1d_minimum():
    s = TPESampler({"x": C.Interval(0.0, 1.0)}, seed=1, n_startup=6)
    for _ in range(40):
        t = s.ask()
        assert 0.0 <= t["x"] <= 1.0
        s.tell(t, (t["x"] - 0.27) ** 2)
    best, val = s.best
    assert abs(best["x"] - 0.27) < 0.05 and val < 2.5e-3
    s.tell({"x": 0.5}, float("nan"))            # failed trials are tolerated
    assert s.ask() is not None


# ------------------------------------------------------------------------------ search rounds / scheduling
def test_search_rounds_keep_the_model_in_the_loop():
    """A sampler learns only between rounds.  Round size = trials_per_gpu x world (round 2) made the shipped study
    (n_trials 15, n_jobs 5, TPE start-up 8) ONE round on three or more GPUs -- a random search.  plan_rounds never
    plans fewer than three rounds, and the reference's own sequence is available as a mode."""
    from utilities.sweep.farm import plan_rounds
    # the shipped sweeper config (conf/experiment/optimization/corner_smoothing.yaml:26-27)
    assert plan_rounds(15, 5, 1) == [5, 5, 5] == plan_rounds(15, 5, 8) == plan_rounds(15, 5, 8, mode="reference")
    # BASELINE config 5: 64 trials, n_jobs 8
    assert plan_rounds(64, 8, 1) == [8] * 8 == plan_rounds(64, 8, 8, mode="reference")      # the reference's asks / tells
    assert plan_rounds(64, 8, 8) == [22, 21, 21]                  # throughput mode: as few rounds as allowed, never < 3
    assert plan_rounds(64, 8, 2) == [16] * 4 and plan_rounds(64, 8, 4) == [22, 21, 21]
    assert plan_rounds(64, 8, 8, per_gpu=1) == [8] * 8            # trials_per_gpu = 1: one trial per GPU and round
    assert plan_rounds(2, 8, 8) == [2] and plan_rounds(0, 8, 8) == []       # never MORE rounds than the reference's own
    assert plan_rounds(8, 4, 1) == [4, 4] == plan_rounds(8, 4, 8)
    assert sum(plan_rounds(37, 5, 3)) == 37
    with pytest.raises(ValueError):
        plan_rounds(10, 2, 1, mode="fastest")
    # at world = 8 with the shipped config some trial must come from the TPE branch
    s = TPESampler({"x": C.Interval(0.01, 0.10)}, seed=0)          # n_startup = 8
    guided = []
    for size in plan_rounds(15, 5, 8):
        guided.append(s.is_guided())
        batch = [s.ask() for _ in range(size)]
        for b in batch:
            s.tell(b, (b["x"] - 0.03) ** 2)
    assert guided == [False, False, True]                          # 10 results told before the third round
    s = TPESampler({"x": C.Interval(0.01, 0.10)}, seed=0)
    guided = []
    for size in plan_rounds(64, 8, 8):
        guided.append(s.is_guided())
        for b in [s.ask() for _ in range(size)]:
            s.tell(b, (b["x"] - 0.03) ** 2)
    assert guided == [False, True, True]


def test_trial_cost_follows_the_measured_iteration_counts():
    """Scheduling weight = measured iterations (N, Re) x measured time per iteration (N), by solver class
    (profiles/r02_sweeps_streams.md); N^5 ignored Re and the solver class."""
    from utilities.sweep.farm import trial_cost, expected_iterations, us_per_iteration
    assert expected_iterations(256, 1000) == 1299041 and expected_iterations(64, 400) == 273012
    assert trial_cost(dict(N=256, Re=1000)) == pytest.approx(1299041 * 35.1e-6)           # a lone trial on the chip-wide kernel (round 4)
    assert trial_cost(dict(N=256, Re=1000)) > trial_cost(dict(N=256, Re=400)) > trial_cost(dict(N=256, Re=100))
    assert trial_cost(dict(N=128, Re=400)) > trial_cost(dict(N=128, Re=1000))            # 895 k vs 833 k iterations
    assert us_per_iteration(512) == pytest.approx(35.1 * 8) and 10.8 <= us_per_iteration(48) <= 16.9
    assert 2.0 < trial_cost(dict(N=48, Re=250)) < trial_cost(dict(N=64, Re=250))        # interpolated, monotone in N
    fsg = trial_cost(dict(N=128, Re=1000), solver="solvers.spectral.fsg.FSGSolver")
    assert fsg < trial_cost(dict(N=128, Re=1000)) and fsg == trial_cost(dict(N=128, Re=1000, solver="spectral/fsg"))


def test_trial_cost_is_what_the_kernels_of_round_3_and_4_measured():
    """An N=64 trial weighs what the one-XCD kernel takes (16.9 us per iteration alone -- the launch-path figure of round 2,
    31.8, made LPT weigh small-N trials 2x too heavy against N=256), a trial inside a batch what a batch takes per trial, and
    config 5 (64 FSG trials at N=128, eight per round) what profiles/r03_sweeps.md measured end to end."""
    from utilities.sweep.farm import trial_cost, us_per_iteration
    assert trial_cost(dict(N=64, Re=100)) == pytest.approx(306441 * 16.9e-6)               # 5.2 s, not 9.7
    assert trial_cost(dict(N=256, Re=100)) / trial_cost(dict(N=64, Re=100)) == pytest.approx(1050762 * 35.1 / (306441 * 16.9))
    assert trial_cost(dict(N=64, Re=100), batch=8) == pytest.approx(306441 * 2.14e-6)      # eight per launch: 467.6 k trial-it/s
    assert trial_cost(dict(N=64, Re=100), batch=64) == trial_cost(dict(N=64, Re=100), batch=8)
    lone, two, eight = (us_per_iteration(128, b) for b in (1, 2, 8))
    assert lone == pytest.approx(24.3) and eight == pytest.approx(7.6) and eight < two < lone
    # config 5: 64 trials in rounds of eight on one GPU took 68 s end to end (records included): ~1.06 s per trial
    fsg8 = trial_cost(dict(N=128, Re=1000), solver="solvers.spectral.fsg.FSGSolver", batch=8)
    assert 0.7 < fsg8 < 1.2
    assert trial_cost(dict(N=128, Re=1000), solver="solvers.spectral.fsg.FSGSolver") == pytest.approx((250_000 * 16.1 + 35_000 * 29.1) * 1e-6)


def test_a_failing_unbatched_trial_costs_only_its_own_record(tmp_path, monkeypatch):
    """run_group: trials that do not share launches (here two sizes with one member each) run one by one; one of
    them raising must not wipe the record of the other (utilities.sweep.farm: 'costs only its own trials')."""
    import importlib.util
    import sys
    spec = importlib.util.spec_from_file_location("ldc_main_under_test", PKG / "main.py")
    main = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(main)

    def fake_run_solver(cfg, out_dir, device=None):
        if cfg["N"] == 24:
            raise RuntimeError("solver blew up")
        return dict(objective=0.5, N=cfg["N"], Re=cfg["Re"])

    monkeypatch.setattr(main, "run_solver", fake_run_solver)
    monkeypatch.chdir(tmp_path)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(main.FarmError) as ei:
        main.main(["-m", "N=16,24,40", "Re=100", f"hydra.sweep.dir={tmp_path}/sweep"])
    recs = ei.value.records
    assert [("error" in r) for r in recs] == [False, True, False]
    assert recs[0]["objective"] == 0.5 and recs[2]["N"] == 40 and "solver blew up" in recs[1]["error"]
    saved = json.loads((tmp_path / "sweep" / "sweep_results.json").read_text())
    assert [("error" in r) for r in saved] == [False, True, False]
