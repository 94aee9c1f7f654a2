"""N > 1 path on CPU: 2 ranks over gloo run the sweep farm, the max-over-ranks reduction that bench.py
uses, and a batched ask/tell optimisation; plus the pure scheduling logic."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

from utilities.sweep.farm import Dist, assign_lpt, run_farm, trial_cost

ROOT = Path(__file__).resolve().parent.parent


def test_lpt_assignment_balances_the_grid_sweep():
    trials = [dict(N=n, Re=re) for n in (64, 128, 256) for re in (100, 400, 1000)]
    costs = [trial_cost(t) for t in trials]
    own = assign_lpt(costs, 8)
    assert sorted(set(own)) == list(range(8))                 # every GPU gets work
    heavy = [own[i] for i, t in enumerate(trials) if t["N"] == 256]
    assert len(set(heavy)) == 3                               # the three N=256 trials sit on three GPUs
    assert assign_lpt(costs, 8) == own                        # deterministic
    assert assign_lpt([5, 4, 3, 2, 1], 2) == [0, 1, 1, 0, 0] or sum(1 for _ in own) == 9


def test_single_process_farm_is_a_plain_loop():
    d = Dist()
    assert d.world == 1
    out = run_farm([dict(N=4), dict(N=8)], lambda t, i: dict(v=t["N"] * 2), d)
    assert [r["v"] for r in out] == [8, 16] and [r["trial_index"] for r in out] == [0, 1]
    assert d.max_float(3.5) == 3.5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_farm(tmp_path):
    out = tmp_path / "farm.json"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           str(ROOT / "tests" / "_farm_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads(out.read_text())
    assert res["world"] == 2 and res["tmax"] == 2.0
    recs = res["recs"]
    assert [(t["N"], t["Re"]) for t in recs] == [(n, re) for n in (64, 128, 256) for re in (100, 400, 1000)]
    assert [t["trial_index"] for t in recs] == list(range(9))
    ranks = {t["rank"] for t in recs}
    assert ranks == {0, 1}
    assert all(t["pid_rank"] == t["rank"] for t in recs)      # each trial ran exactly where it was assigned
    # both ranks hold the same sampler state after the gathered tells
    r0, r1 = (json.loads((tmp_path / f"rank{k}.json").read_text()) for k in (0, 1))
    assert r0 == r1 and r0["n"] == 12
    assert res["best"] <= min(h for hh in res["hist"][:2] for h in hh)
    # search round of the launcher: 4 candidates per GPU x 2 ranks, each rank ran ONE group of 4 equal-N trials
    g0, g1 = (json.loads((tmp_path / f"groups{k}.json").read_text()) for k in (0, 1))
    assert len(g0["groups"]) == 1 and len(g1["groups"]) == 1
    assert len(g0["groups"][0]) == 4 and len(g1["groups"][0]) == 4
    assert sorted(g0["groups"][0] + g1["groups"][0]) == list(range(8))
    assert [r["batch_size"] for r in res["round_recs"]] == [4] * 8
    assert {r["rank"] for r in res["round_recs"]} == {0, 1}
    # the failing trial: FarmError on BOTH ranks, the three finished trials kept
    for g in (g0, g1):
        fe = g["farm_error"]
        assert fe is not None and "boom" in fe["msg"]
        assert fe["errors"] == [False, True, False, False]
        assert fe["kept"][0] == 0.0 and fe["kept"][2] == 2.0 and fe["kept"][3] == 3.0


def test_farm_groups_equal_n_trials_for_batching():
    d = Dist()
    trials = [dict(N=16, Re=r) for r in (1, 2, 3)] + [dict(N=32, Re=9)]
    seen = []

    def run_group(items):
        seen.append([i for i, _ in items])
        return [dict(v=t["Re"]) for _, t in items]

    out = run_farm(trials, None, d, run_group=run_group, group_key=lambda t: t["N"])
    assert seen == [[0, 1, 2], [3]]
    assert [r["v"] for r in out] == [1, 2, 3, 9] and [r["batch_size"] for r in out] == [3, 3, 3, 1]


def test_tpe_treats_a_nan_objective_as_a_failed_trial():
    """A diverged trial has psi_min = NaN, so its botella_vortex objective is NaN; it must neither be
    reported as the optimum nor poison the Parzen windows."""
    import math
    from utilities.config.compose import Interval
    from utilities.sweep.farm import TPESampler
    s = TPESampler({"a": Interval(0.0, 1.0)}, seed=0)
    s.tell({"a": 0.1}, 1.0)
    s.tell({"a": 0.2}, float("nan"))
    s.tell({"a": 0.3}, 0.5)
    s.tell({"a": 0.4}, None)
    best, val = s.best
    assert best == {"a": 0.3} and val == 0.5
    assert s.values[1] == math.inf and s.values[3] == math.inf
    only_bad = TPESampler({"a": Interval(0.0, 1.0)}, seed=0)
    only_bad.tell({"a": 0.2}, float("nan"))
    assert only_bad.best[1] == math.inf
    assert 0.0 <= only_bad.ask()["a"] <= 1.0


def test_farm_error_keeps_finished_trials_single_process():
    from utilities.sweep.farm import FarmError

    def run(t, i):
        if t["N"] == 8:
            raise ValueError("bad config")
        return dict(v=t["N"])

    with pytest.raises(FarmError) as ei:
        run_farm([dict(N=4), dict(N=8), dict(N=16)], run, Dist())
    recs = ei.value.records
    assert recs[0]["v"] == 4 and recs[2]["v"] == 16 and "bad config" in recs[1]["error"]
    out = run_farm([dict(N=4), dict(N=8)], run, Dist(), raise_on_error=False)
    assert out[0]["v"] == 4 and out[1]["objective"] == float("inf")
