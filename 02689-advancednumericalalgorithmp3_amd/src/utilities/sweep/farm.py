"""Sweep farm: one independent trial at a time per GPU, host-side gather.

The reference parallelises over trials only (joblib launcher / Optuna n_jobs / LSF arrays,
conf/machine/local.yaml:5-9, scripts/hpc_submit.py:103-107); a trajectory itself is sequential.
Here the workers are the ranks of ``torch.distributed.run`` (one per MI355X, backend nccl = RCCL;
gloo on CPU for tests).  There is NO collective in the data path: each rank runs its trials on its
own GPU and only the small result records are gathered (``all_gather_object``).
"""
from __future__ import annotations

import logging
import math
import os
import time

import numpy as np


log = logging.getLogger(__name__)


# Measured on one MI355X.  Iterations to the reference's convergence criterion: BASELINE config 4 at full size
# (profiles/r02_sweeps_streams.md).  Microseconds per full iteration (with E / Z / P) by the kernel a trial really gets:
#   * a LONE trial: the one-XCD kernel up to N = 79 (profiles/r03_xcd_ab.log: N=16 10.8, 32 12.5, 64 16.9 -- the launch path it
#     replaced took 28.6 / 29.6 / 30.6), the chip-wide kernel from N = 80 (profiles/r04_wide_ab_*.log: N=128 24.3, 256 35.1;
#     launch path 35.7 / 50.2);
#   * a trial INSIDE a batch of equal-N trials (>= 8 of them; 1e6 / trial-iterations per second): N=16 and 32 on the
#     trial-per-CU / one-XCD kernels (profiles/r03_cu_ab.log: 64 trials 5.7 M/s and 3.2 M/s), N=64 eight per launch on the
#     one-XCD kernel (r03_xcd_ab.log: 467.6 k/s), N=128 two halves of eight on two streams of the launch path (BENCH_r03 farm:
#     131.3 k/s), N=256 two trials on two streams (23.4 k/s).
# The smoother (FSG levels: no E / Z / P, the pressure transformed every stage): r03_xcd_ab.log, r04_wide_ab_*.log.
_SG_ITERATIONS = {(64, 100): 306441, (64, 400): 273012, (64, 1000): 247661,
                  (128, 100): 656077, (128, 400): 895532, (128, 1000): 832853,
                  (256, 100): 1050762, (256, 400): 1088448, (256, 1000): 1299041}
_US_PER_ITERATION = {16: 10.8, 32: 12.5, 64: 16.9, 128: 24.3, 256: 35.1}               # SG, lone trial
_US_PER_ITERATION_BATCHED = {16: 0.175, 32: 0.31, 64: 2.14, 128: 7.6, 256: 42.7}       # SG, per trial of a batch
_US_PER_SMOOTHER_ITERATION = {16: 10.5, 32: 11.7, 64: 16.1, 128: 29.1, 256: 71.4}      # FSG level, lone (N=256: launch path)
_US_PER_SMOOTHER_ITERATION_BATCHED = {16: 0.32, 32: 0.81, 64: 2.14, 128: 12.2, 256: 68.0}
_BATCH_FULL = 8                       # from this many equal-N trials on a GPU the batched figure applies; between 1 and it: linear in 1/B
_FSG_COARSE_ITERATIONS = 250_000      # config 5 (N=128, Re=1000): the 64-level takes ~250 k iterations, the 128-level ~35 k
_FSG_FINE_ITERATIONS = 35_000         # (profiles/r03_sweeps.md: 64 trials, 18.36 M trial-iterations, 68 s on one GPU)


def _interp_log(table: dict, x: float) -> float:
    """Piecewise log-log interpolation of a {size: value} table, power-law extrapolation beyond its ends."""
    keys = sorted(table)
    if x <= keys[0]:
        lo, hi = keys[0], keys[1]
    elif x >= keys[-1]:
        lo, hi = keys[-2], keys[-1]
    else:
        hi = next(k for k in keys if k >= x)
        lo = keys[keys.index(hi) - 1] if hi != x else hi
        if lo == hi:
            return float(table[hi])
    slope = math.log(table[hi] / table[lo]) / math.log(hi / lo)
    return float(table[lo] * (x / lo) ** slope)


def us_per_iteration(n: float, batch: int = 1, smoother: bool = False) -> float:
    """Microseconds per iteration of one SG trial of order n (beyond N=256 a stage is MFMA-bound: ~N^3) that shares its
    GPU with ``batch`` - 1 other trials of the same order; ``smoother``: an FSG level."""
    lone, full = ((_US_PER_SMOOTHER_ITERATION, _US_PER_SMOOTHER_ITERATION_BATCHED) if smoother
                  else (_US_PER_ITERATION, _US_PER_ITERATION_BATCHED))

    def at(table):
        if n > 256:
            return table[256] * (n / 256.0) ** 3
        return _interp_log(table, max(n, 16.0))

    b = max(1, min(int(batch), _BATCH_FULL))
    if b == 1:
        return at(lone)
    w = (1.0 - 1.0 / b) / (1.0 - 1.0 / _BATCH_FULL)          # 0 for a lone trial, 1 from _BATCH_FULL trials
    return (1.0 - w) * at(lone) + w * at(full)


def expected_iterations(n: float, re: float) -> float:
    """Iterations to the reference's stopping rule: the measured table at the nearest Re, log-log in N."""
    res = sorted({r for _, r in _SG_ITERATIONS})
    r = min(res, key=lambda q: abs(math.log(q / max(re, 1e-9))))
    return _interp_log({k: v for (k, rr), v in _SG_ITERATIONS.items() if rr == r}, max(n, 8.0))


def trial_cost(trial: dict, solver: str = None, batch: int = 1) -> float:
    """Expected GPU seconds of a trial, for longest-first scheduling: measured iteration counts (N, Re) times the
    measured time per iteration (N, and how many equal-N trials share the GPU: ``batch``), by solver class -- an FSG
    trial is its coarse level plus a short fine level in smoother mode.  (Round 2 used N^5, which ignores Re and the
    solver class; round 3 launch-path times only, which weighed an N=64 trial 2x too heavy against N=256.)
    ``solver``: class hint ("fsg" in it selects the FSG model); a trial's own "solver" entry wins."""
    t = dict(trial)
    n, re = float(t.get("N", 32)), float(t.get("Re", 100))
    kind = str(t.get("solver", solver or "")).lower()
    if "fsg" in kind:
        levels = int(t.get("n_levels", t.get("solver.n_levels", 2)))
        cost, m = 0.0, n
        for lvl in range(levels):
            its = _FSG_FINE_ITERATIONS if lvl == 0 and levels > 1 else _FSG_COARSE_ITERATIONS
            cost += its * us_per_iteration(m, batch, smoother=True)
            if m // 2 < 12:
                break
            m //= 2
        return cost * 1e-6
    return expected_iterations(n, re) * us_per_iteration(n, batch) * 1e-6


def plan_rounds(n_trials: int, n_jobs: int, world: int, per_gpu: int = None, mode: str = "throughput",
                min_rounds: int = 3) -> list:
    """Sizes of the ask/tell rounds of a model-based search (sum = n_trials).

    The reference asks Optuna for ``n_jobs`` candidates, runs them as ``n_jobs`` processes and tells the results
    (conf/experiment/optimization/corner_smoothing.yaml:50-57; scripts/hpc_submit.py:103-107 for grids).  A sampler
    only learns BETWEEN rounds, so the round size is a property of the search, not only of the machine:

    * ``mode="reference"``: every round is ``n_jobs`` candidates whatever the world size -- the reference's own
      sequence of asks and tells.  More GPUs then only spread a round (n_jobs = 8 on 8 GPUs: one trial per GPU);
      since one MI355X advances eight equal-N trials as a batch in about the time of one, the wall time of a round
      barely changes: no throughput scaling beyond batching.
    * ``mode="throughput"`` (default): a round offers every GPU ``per_gpu`` candidates (default ``n_jobs``), i.e. up to
      ``per_gpu x world`` -- but never so many that the study has fewer than ``min_rounds`` rounds (default 3; fewer
      only where the reference's own sequence has fewer): with ONE round of n_trials candidates the sampler would
      never see a result and the study would be a random search (round 2's behaviour at world >= n_trials / n_jobs).
      The trials are split evenly over max(min(min_rounds, reference rounds), ceil(n_trials / capacity)) rounds.
      With one GPU (and per_gpu = n_jobs) this IS the reference's sequence; with more GPUs the speed-up of a search is bounded by
      rounds(1 GPU) / rounds(N GPUs) -- config 5 (64 trials, n_jobs 8): 8 rounds -> 3, at most 2.7x at any GPU count.
    """
    n_trials, n_jobs, world = int(n_trials), max(1, int(n_jobs)), max(1, int(world))
    if n_trials <= 0:
        return []
    if mode == "reference":
        size = n_jobs
        return [min(size, n_trials - lo) for lo in range(0, n_trials, size)]
    if mode != "throughput":
        raise ValueError(f"unknown search mode {mode!r}: use 'reference' or 'throughput'")
    capacity = max(1, int(per_gpu if per_gpu is not None else n_jobs)) * world
    ref_rounds = -(-n_trials // n_jobs)                       # what the reference's own sequence has
    n_rounds = min(n_trials, max(min(int(min_rounds), ref_rounds), -(-n_trials // capacity)))
    base, extra = divmod(n_trials, n_rounds)
    return [base + (1 if k < extra else 0) for k in range(n_rounds)]


def assign_lpt(costs, n_workers: int) -> list:
    """Longest-processing-time-first: returns worker index per job (deterministic)."""
    load = [0.0] * n_workers
    owner = [0] * len(costs)
    for j in sorted(range(len(costs)), key=lambda q: (-costs[q], q)):
        w = min(range(n_workers), key=lambda q: (load[q], q))
        owner[j] = w
        load[w] += costs[j]
    return owner


class Dist:
    """Thin view of torch.distributed that also works without it (world size 1)."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self._pg = None

    def init(self, backend: str = None):
        if self.world > 1 and self._pg is None:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:
                # LDC_DIST_BACKEND=gloo: several ranks on ONE card (RCCL wants a GPU per rank) -- the farm gathers
                # host objects only, so gloo carries it just as well; used by the two-rank GPU test
                backend = os.environ.get("LDC_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kw["device_id"] = torch.device(f"cuda:{self.local_rank}")
            dist.init_process_group(backend, **kw)
            self._pg = dist
        return self

    def barrier(self):
        if self._pg is not None:
            self._pg.barrier()

    def all_gather_object(self, obj) -> list:
        if self._pg is None:
            return [obj]
        out = [None] * self.world
        self._pg.all_gather_object(out, obj)
        return out

    def max_float(self, x: float) -> float:
        """max over ranks of a host scalar (used for the bench's max-over-ranks time)."""
        return max(self.all_gather_object(float(x)))

    def close(self):
        if self._pg is not None:
            self._pg.destroy_process_group()
            self._pg = None


def run_farm(trials: list, run_trial, dist: Dist, cost=trial_cost, run_group=None, group_key=None,
             raise_on_error: bool = True) -> list:
    """Run every trial exactly once; returns the list of result records on every rank, in trial order.

    ``run_trial(trial, index)`` -> JSON-able dict.  Scheduling is static LPT computed identically on all
    ranks, so no work needs to be communicated.  With ``run_group`` the trials a rank owns are grouped by
    ``group_key(trial)`` and handed over together -- ``run_group([(index, trial), ...]) -> [record, ...]`` --
    so that equal-N trials can share every kernel launch on that GPU (solvers.spectral.batched).

    A trial (or group) that raises does not take the sweep down with it: its records become
    ``{"error": repr(exc), "objective": inf}``, every rank still reaches the gather, and afterwards
    ``FarmError`` (carrying all records) is raised on all ranks alike unless ``raise_on_error`` is false --
    like the reference's joblib / Optuna launchers, which surface the exception and keep finished trials."""
    # how many trials of the same order a GPU will see: they share their launches there, and a trial inside a batch costs a
    # fraction of a lone one (N=64: 2.1 us per iteration instead of 16.9) -- passed to cost functions that take it
    census = {}
    for t in trials:
        census[t.get("N")] = census.get(t.get("N"), 0) + 1
    try:
        costs = [cost(t, batch=max(1, census[t.get("N")] // dist.world)) for t in trials]
    except TypeError:          # a caller's own cost(trial)
        costs = [cost(t) for t in trials]
    owner = assign_lpt(costs, dist.world)
    mine = {}
    my = [(idx, t) for idx, t in enumerate(trials) if owner[idx] == dist.rank]
    if run_group is None:
        groups = [[it] for it in my]
    else:
        by_key = {}
        for it in my:
            by_key.setdefault(group_key(it[1]) if group_key else None, []).append(it)
        groups = list(by_key.values())
    for grp in groups:
        t0 = time.perf_counter()
        try:
            recs = run_group(grp) if run_group is not None else [run_trial(grp[0][1], grp[0][0])]
        except Exception as exc:          # keep going: every rank must reach the gather below
            log.exception("trial group %s failed on rank %d", [i for i, _ in grp], dist.rank)
            recs = [dict(error=repr(exc), objective=math.inf) for _ in grp]
        dt = time.perf_counter() - t0
        for (idx, _), rec in zip(grp, recs):
            rec = dict(rec)
            # a group is handed over and comes back as a whole: it has ONE wall time (batch_seconds);
            # trial_seconds is the amortised time per trial
            rec.update(trial_index=idx, rank=dist.rank, batch_seconds=dt, batch_size=len(grp),
                       trial_seconds=dt / len(grp))
            mine[idx] = rec
    merged = {}
    for part in dist.all_gather_object(mine):
        merged.update(part)
    out = [merged[i] for i in range(len(trials))]
    failed = [r["trial_index"] for r in out if "error" in r]
    if failed and raise_on_error:
        raise FarmError(f"trials {failed} failed: " + "; ".join(out[i]["error"] for i in failed[:3]), out)
    return out


class FarmError(RuntimeError):
    """Raised on EVERY rank after the gather when any trial failed; ``records`` holds all trial records
    (finished ones intact, failed ones as ``{"error": ..., "objective": inf}``)."""

    def __init__(self, msg, records):
        super().__init__(msg)
        self.records = records


class TPESampler:
    """Small tree-structured Parzen estimator for low-dimensional mixed spaces (minimisation).

    Stands in for Optuna's TPE sampler behind ``conf/hydra/sweeper/optuna_corner.yaml`` (Optuna is not
    installed).  Deterministic for a given seed; every rank holds an identical copy and is told the
    same gathered results, so no sampler state is communicated."""

    def __init__(self, space: dict, seed: int = 0, n_startup: int = 8, gamma: float = 0.25, n_candidates: int = 24):
        from utilities.config.compose import Interval
        self.space, self.rng = dict(space), np.random.default_rng(seed)
        self.n_startup, self.gamma, self.n_candidates = n_startup, gamma, n_candidates
        self.Interval = Interval
        self.trials, self.values = [], []

    def _random(self):
        out = {}
        for k, dom in self.space.items():
            if isinstance(dom, self.Interval):
                out[k] = float(self.rng.uniform(dom.low, dom.high))
            else:
                out[k] = dom[int(self.rng.integers(len(dom)))]
        return out

    def _score(self, x, group):
        """log density of x under a Parzen window over `group` (product over dimensions)."""
        s = 0.0
        for k, dom in self.space.items():
            vals = [g[k] for g in group]
            if isinstance(dom, self.Interval):
                width = dom.high - dom.low
                bw = max(width / max(len(vals), 1) ** 0.5 / 2.0, 1e-3 * width)
                z = (x[k] - np.array(vals)) / bw
                s += math.log(np.mean(np.exp(-0.5 * z * z)) / bw + 1e-300)
            else:
                cnt = sum(1 for v in vals if v == x[k])
                s += math.log((cnt + 1.0) / (len(vals) + len(dom)))
        return s

    def is_guided(self) -> bool:
        """True once ``ask`` draws from the model (enough finite results have been told), False while it samples
        the prior."""
        return sum(1 for v in self.values if math.isfinite(v)) >= self.n_startup

    def ask(self) -> dict:
        finite = [(t, v) for t, v in zip(self.trials, self.values) if math.isfinite(v)]
        if len(finite) < self.n_startup:
            return self._random()
        finite.sort(key=lambda tv: tv[1])
        n_good = max(1, int(math.ceil(self.gamma * len(finite))))
        good, bad = [t for t, _ in finite[:n_good]], [t for t, _ in finite[n_good:]] or [t for t, _ in finite]
        best, best_s = None, -math.inf
        for _ in range(self.n_candidates):
            base = good[int(self.rng.integers(len(good)))]
            cand = {}
            for k, dom in self.space.items():
                if isinstance(dom, self.Interval):
                    width = dom.high - dom.low
                    x = base[k] + self.rng.normal(0.0, width / 6.0)
                    while x < dom.low or x > dom.high:          # reflect at the bounds (clipping piles up there)
                        x = 2 * dom.low - x if x < dom.low else 2 * dom.high - x
                    cand[k] = float(x)
                else:
                    cand[k] = base[k] if self.rng.random() < 0.7 else dom[int(self.rng.integers(len(dom)))]
            sc = self._score(cand, good) - self._score(cand, bad)
            if sc > best_s:
                best, best_s = cand, sc
        return best

    def tell(self, trial: dict, value: float):
        """A non-finite objective (a diverged trial gives NaN) counts as a failed trial: +inf, like Optuna."""
        v = math.inf if value is None else float(value)
        self.trials.append(dict(trial))
        self.values.append(v if math.isfinite(v) else math.inf)

    @property
    def best(self):
        if not self.values:
            return None, math.inf
        i = int(np.argmin(self.values))          # values are finite or +inf (tell): never NaN
        return self.trials[i], self.values[i]
