"""Worker for tests/test_farm_gloo.py: launched by torch.distributed.run with 2 ranks (gloo, CPU)."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src"))
from utilities.config.compose import Interval  # noqa: E402
from utilities.sweep.farm import Dist, FarmError, TPESampler, run_farm  # noqa: E402

out = Path(sys.argv[1])
dist = Dist().init("gloo")
trials = [dict(N=n, Re=re) for n in (64, 128, 256) for re in (100, 400, 1000)]     # BASELINE config 4


def fake_trial(t, idx):
    time.sleep(0.01)
    return dict(N=t["N"], Re=t["Re"], objective=(t["N"] - 100) ** 2 * 1e-4 + t["Re"] * 1e-6, pid_rank=dist.rank)


recs = run_farm(trials, fake_trial, dist)
# the timing reduction bench.py uses: max over ranks of a host scalar
tmax = dist.max_float(1.0 + dist.rank)
# an optimisation loop: identical sampler copies on all ranks, results all-gathered
sampler = TPESampler({"x": Interval(0.0, 1.0), "N": [30, 40, 50]}, seed=0, n_startup=4)
hist = []
for _ in range(6):
    batch = [sampler.ask() for _ in range(dist.world)]
    res = run_farm(batch, lambda b, i: dict(objective=(b["x"] - 0.3) ** 2 + (b["N"] - 40) ** 2 * 1e-3), dist,
                   cost=lambda b: 1.0)
    for b, r in zip(batch, res):
        sampler.tell(b, r["objective"])
    hist.append([r["objective"] for r in res])
# the launcher's search round (main.py): trials_per_gpu x world candidates per round, equal-N trials handed to a
# rank TOGETHER so that they share its launches -- every rank must get a whole group, not one trial
per_gpu = 4
groups_seen = []


def run_group(items):
    groups_seen.append(sorted(i for i, _ in items))
    return [dict(objective=(t["x"] - 0.5) ** 2, N=t["N"]) for _, t in items]


s2 = TPESampler({"x": Interval(0.0, 1.0)}, seed=1, n_startup=4)
round_batch = [dict(s2.ask(), N=128) for _ in range(per_gpu * dist.world)]
round_recs = run_farm(round_batch, None, dist, run_group=run_group, group_key=lambda t: t["N"])

# a failing trial must not strand the other ranks: everyone reaches the gather, FarmError everywhere
def flaky(t, idx):
    if idx == 1:
        raise RuntimeError("boom")
    return dict(objective=float(idx))


try:
    run_farm([dict(N=8), dict(N=8), dict(N=8), dict(N=8)], flaky, dist, cost=lambda t: 1.0)
    farm_error = None
except FarmError as exc:
    farm_error = dict(msg=str(exc), kept=[r.get("objective") for r in exc.records],
                      errors=[("error" in r) for r in exc.records])
dist.barrier()
if dist.rank == 0:
    out.write_text(json.dumps(dict(recs=recs, tmax=tmax, world=dist.world, best=sampler.best[1], hist=hist,
                                   round_recs=round_recs)))
(out.parent / f"groups{dist.rank}.json").write_text(json.dumps(dict(groups=groups_seen, farm_error=farm_error)))
(out.parent / f"rank{dist.rank}.json").write_text(json.dumps(dict(best=sampler.best[1], n=len(sampler.values))))
dist.close()
