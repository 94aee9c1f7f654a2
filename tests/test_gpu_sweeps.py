"""BASELINE configs 4 and 5 THROUGH the launcher (main.py), checked against the reference's fixtures and the oracle.

Config 4: ``-m N=64,128,256 Re=100,400,1000`` -- a grid sweep whose equal-N trials share their launches on a GPU
(solvers.spectral.batched).  Config 5: the corner-smoothing search on ``spectral/fsg`` with the botella_vortex
objective (conf/experiment/optimization/corner_smoothing.yaml:25,50-57) -- ask/tell rounds of batched FSG trials.
The full-size runs take minutes to hours; here the same command lines run with iteration caps, and what they
write (``results.json``, ``solution.vts``, ``sweep_results.json``) is compared with the reference's own
trajectories (tests/golden/g4_*), with oracle runs on identical parameters, and with stand-alone solves."""
import importlib.util
import json

import numpy as np
import pytest

from oracle import ldc_oracle as orc
from conftest import PKG

pytestmark = pytest.mark.gpu


@pytest.fixture()
def launcher(tmp_path, monkeypatch):
    spec = importlib.util.spec_from_file_location("ldc_main_sweeps", PKG / "main.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.chdir(tmp_path)

    def run(argv):
        before = set(tmp_path.glob("hydra_outputs/multirun/*/*"))
        out = mod.main(argv)
        new = sorted(set(tmp_path.glob("hydra_outputs/multirun/*/*")) - before)
        assert len(new) == 1, new
        return out, new[0]

    return run


def _fields(run_dir, M):
    """u, v of solution.vts as [ix, iy] arrays (the file is y-slow / x-fast, reference base.py:464-549)."""
    from solvers.vtkio import read_vts
    g = read_vts(run_dir / "solution.vts")
    return g["point_data"]["u"].reshape(M, M).T, g["point_data"]["v"].reshape(M, M).T


def test_config4_batched_sweep_vs_reference_and_oracle(launcher, golden_dir):
    """``-m N=64 Re=400,1000 max_iterations=1000``: two N=64 trials advanced by the same launches.  Re=400 after 1000
    iterations is the reference's own trajectory fixture; Re=1000 is checked against the oracle."""
    _, root = launcher(["-m", "N=64", "Re=400,1000", "max_iterations=1000"])
    recs = json.loads((root / "sweep_results.json").read_text())
    assert [(r["N"], r["Re"]) for r in recs] == [(64, 400), (64, 1000)]
    assert all(r["metrics"]["iterations"] == 1000 and not r["metrics"]["converged"] for r in recs)
    # one share group of two on this rank: N=64 runs on the small-N trial kernel, both trials in ONE launch (an XCD each)
    assert all(r["solve_group_size"] == 2 and r["solve_streams"] == 1 and r["solve_batch_size"] == 2 for r in recs)
    g = np.load(golden_dir / "g4_traj_N64_Re400_K1000.npz")
    u, v = _fields(root / "0", 65)
    assert np.max(np.abs(u - g["u"].reshape(65, 65))) < 1e-12
    assert np.max(np.abs(v - g["v"].reshape(65, 65))) < 1e-12
    m = recs[0]["metrics"]
    assert m["final_energy"] == pytest.approx(g["E"][-1], rel=1e-10)
    assert m["final_enstrophy"] == pytest.approx(g["Z"][-1], rel=1e-10)
    assert m["final_palinstrophy"] == pytest.approx(g["P"][-1], rel=1e-10)
    assert m["u_momentum_residual"] == pytest.approx(g["res"][-1, 0], rel=1e-10)
    vm = dict(zip((str(k) for k in g["vortex_keys"]), g["vortex_vals"]))
    for key in ("psi_min", "psi_min_x", "psi_min_y", "omega_center", "omega_max"):
        assert m[key] == pytest.approx(vm[key], rel=1e-9, abs=1e-9), key
    o = orc.OracleSG(64, 1000.0)
    for _ in range(1000):
        o.step()
    u, v = _fields(root / "1", 65)
    assert np.max(np.abs(u - o.u)) < 1e-12 and np.max(np.abs(v - o.v)) < 1e-12
    assert recs[1]["metrics"]["final_energy"] == pytest.approx(o.energy(), rel=1e-10)


def test_config4_grid_shape_n64_to_n256(launcher):
    """The literal config-4 grid, ``-m N=64,128,256 Re=100,400,1000``, capped at 40 iterations: nine trials in three
    batches (one per N); one trial of each size against the oracle."""
    _, root = launcher(["-m", "N=64,128,256", "Re=100,400,1000", "max_iterations=40"])
    recs = json.loads((root / "sweep_results.json").read_text())
    assert [(r["N"], r["Re"]) for r in recs] == [(n, re) for n in (64, 128, 256) for re in (100, 400, 1000)]
    assert all(r["metrics"]["iterations"] == 40 and r["solve_group_size"] == 3 and r["batch_size"] == 9 for r in recs)
    # N=64 and N=128: a batch of one and a batch of two each; N=256 (a trial fills the chip alone): one by one
    # N=64: one batch of three on the small-N kernel; N=128: two batches (1 + 2) on two streams; N=256: one by one
    assert [r["solve_batch_size"] for r in recs] == [3, 3, 3, 1, 2, 2, 1, 1, 1]
    for idx in (1, 5, 6):           # (64, 400), (128, 1000), (256, 100)
        N, Re = recs[idx]["N"], recs[idx]["Re"]
        o = orc.OracleSG(N, float(Re))
        for _ in range(40):
            o.step()
        u, v = _fields(root / str(idx), N + 1)
        assert np.max(np.abs(u - o.u)) < 1e-12 and np.max(np.abs(v - o.v)) < 1e-12, (N, Re)
        assert recs[idx]["metrics"]["final_energy"] == pytest.approx(o.energy(), rel=1e-10)
        assert recs[idx]["metrics"]["final_enstrophy"] == pytest.approx(o.enstrophy(), rel=1e-9)


def test_config5_corner_smoothing_search_vs_oracle(launcher):
    """``+experiment/optimization=corner_smoothing`` with the botella_vortex objective: 8 trials in rounds of 4
    batched FSG solves (N=32: levels 16 -> 32, 300 iterations per level).  Trial 0 against the oracle's FSG driver
    on the corner_smoothing the sampler drew; every trial's objective recomputed from its own vortex metrics."""
    from types import SimpleNamespace
    from solvers import validation as V
    best, root = launcher(["+experiment/optimization=corner_smoothing", "N=32", "hydra.sweeper.n_trials=8",
                           "hydra.sweeper.n_jobs=4", "max_iterations=300", "optuna.objective=botella_vortex"])
    recs = json.loads((root / "sweep_results.json").read_text())
    assert len(recs) == 8 and [r["trial_index"] for r in recs] == [0, 1, 2, 3, 0, 1, 2, 3]
    # (levels 16 -> 32 both run on the small-N trial kernel: the four trials of a round are ONE batch, an XCD slot each)
    assert all(r["solve_group_size"] == 4 and r["solve_batch_size"] == 4 and r["solver"] == "spectral_fsg" for r in recs)
    cs = [r["overrides"]["solver.corner_smoothing"] for r in recs]
    assert all(0.01 <= c <= 0.10 for c in cs) and len(set(cs)) == 8
    assert all(r["params"]["corner_smoothing"] == c for r, c in zip(recs, cs))
    assert all(r["metrics"]["iterations"] == 600 for r in recs)                  # both levels ran to the cap
    lvl, total, conv = orc.oracle_fsg(32, 1000.0, max_iterations=300, corner_smoothing=cs[0])
    assert total == 600 and not conv
    u, v = _fields(root / "0", 33)
    assert np.max(np.abs(u - lvl.u)) < 1e-10 and np.max(np.abs(v - lvl.v)) < 1e-10
    assert recs[0]["metrics"]["final_energy"] == pytest.approx(lvl.energy(), rel=1e-9)
    objs = [r["objective"] for r in recs]
    assert all(isinstance(x, float) and np.isfinite(x) for x in objs) and best == pytest.approx(min(objs))
    for r in recs:
        m = SimpleNamespace(**r["metrics"])
        assert r["objective"] == pytest.approx(V.compute_botella_vortex_objective(m, 1000), rel=1e-12)
        assert r["objective_kind"] == "botella_vortex"


def test_two_rank_farm_on_the_gpu(tmp_path):
    """The N > 1 path on real kernels: two ranks of torch.distributed.run (gloo for the gather -- RCCL needs a GPU per
    rank and the test box has one; both ranks compute on cuda:0) run a 3 x 2 grid sweep through main.py.  Every rank
    advances ITS equal-N trials as one batch, the gathered records are complete, in grid order, and equal a
    single-process run of the same sweep bit for bit.  Ranks that share a card take the launch path (co-resident launches of
    two PROCESSES cannot be kept apart by a lock in one of them: main.py), so the single process is pinned to it too
    (LDC_PIN_MODE=0) -- every record says which kernel advanced it."""
    import os
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    argv = ["-m", "N=32,48", "Re=100,400,250,50", "max_iterations=120"]
    env = dict(os.environ, LDC_DIST_BACKEND="gloo", OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    two = tmp_path / "two"; one = tmp_path / "one"
    two.mkdir(); one.mkdir()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(PKG / "main.py")] + argv
    r = subprocess.run(cmd, cwd=two, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r1 = subprocess.run([sys.executable, str(PKG / "main.py")] + argv, cwd=one, env=dict(env, LDC_PIN_MODE="0"),
                        capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-3000:]
    load = lambda d: json.loads(next(d.glob("hydra_outputs/multirun/*/*/sweep_results.json")).read_text())   # noqa: E731
    a, b = load(two), load(one)
    assert [(x["N"], x["Re"]) for x in a] == [(n, re) for n in (32, 48) for re in (100, 400, 250, 50)]
    assert {x["rank"] for x in a} == {0, 1} and all(x["solve_group_size"] == 2 for x in a)     # two of either size per rank
    assert all(y["solve_group_size"] == 4 for y in b)           # one process: four per size (launch path: two halves on two streams)
    for x, y in zip(a, b):
        assert x["kernel_mode"] == y["kernel_mode"] == 0
        assert x["metrics"]["iterations"] == y["metrics"]["iterations"] == 120
        for key in ("final_energy", "final_enstrophy", "final_palinstrophy", "u_momentum_residual", "psi_min"):
            assert x["metrics"][key] == y["metrics"][key], key
    root = next(two.glob("hydra_outputs/multirun/*/*"))
    assert sorted(p.name for p in root.iterdir() if p.is_dir()) == [str(k) for k in range(8)]


@pytest.mark.parametrize("streams", [2, 1])
def test_a_failing_batch_costs_only_its_own_trials(tmp_path, monkeypatch, streams):
    """run_batches: two share groups through the pool of worker streams; one batch cannot even be built (a basis the
    solver does not have) -- its trials come back as error records, every other batch is solved and recorded."""
    spec = importlib.util.spec_from_file_location("ldc_main_pool", PKG / "main.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.chdir(PKG)               # data/validation is looked up relative to the package
    monkeypatch.setenv("LDC_BATCH_STREAMS", str(streams))     # 1: every group one batch, one after the other, caller's stream

    def cfg(N, Re, **kw):
        node = dict(_target_="solvers.spectral.sg.SGSolver", name="spectral", Re=float(Re), nx=N, ny=N, tolerance=1e-6,
                    max_iterations=150, check_every=64, graph_iters=8)
        node.update(kw)
        return {"N": N, "Re": Re, "solver": node}

    # (persistent=0: on the launch path a share group is cut into two batches; on the small-N kernel it would be one)
    g32 = [cfg(32, 100, persistent=0), cfg(32, 400, persistent=0), cfg(32, 250, basis_type="fourier", persistent=0), cfg(32, 50, persistent=0)]
    g48 = [cfg(48, 100, persistent=0), cfg(48, 400, persistent=0)]
    out = mod.run_batches([(g32, [tmp_path / f"a{k}" for k in range(4)]), (g48, [tmp_path / f"b{k}" for k in range(2)])])
    assert [len(x) for x in out] == [4, 2]
    bad = [False, False, True, True] if streams == 2 else [True] * 4       # the batch that holds the fourier trial
    assert ["error" in r for r in out[0]] == bad and not any("error" in r for r in out[1])
    assert all("fourier" in r["error"].lower() or "basis" in r["error"].lower() for r in out[0] if "error" in r)
    for r in [x for x in out[0] if "error" not in x] + out[1]:
        assert r["metrics"]["iterations"] == 150 and r["solve_streams"] == streams
    from solvers.spectral.sg import SGSolver
    one = SGSolver(**{k: v for k, v in g48[1]["solver"].items() if k != "_target_"})
    one.solve()
    assert out[1][1]["metrics"]["final_energy"] == one.metrics.final_energy       # the pooled trial is the stand-alone trial
    one.close()
