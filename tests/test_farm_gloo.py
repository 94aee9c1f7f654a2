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
