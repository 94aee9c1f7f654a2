"""Solver framework: the iteration driver, result storage and validation hooks.

Contract: reference src/solvers/base.py -- ``LidDrivenCavitySolver.solve`` (:202-330),
``_store_results`` (:112-200), the public attributes ``params / metrics / fields /
time_series`` used by main.py:93-119.  The reference advances one Python-level ``step()``
per iteration and reduces four norms on the host each time; here a subclass enqueues
``check_every`` iterations of device work at once (hipGraph replays) and the host only
polls a device-side latch, yet iteration counts and histories are identical because the
convergence test itself runs on the device after every iteration.
"""
from __future__ import annotations

import logging
import time
from abc import ABC, abstractmethod

import numpy as np

from .datastructures import Fields, Metrics, TimeSeries, _VORTEX_KEYS

log = logging.getLogger(__name__)

# columns of the per-iteration record block returned by _advance (see include/ldc_hip.h)
REL, RU, RV, RP, EN, ZN, PN, DT = range(8)
WARMUP_ITERATIONS = 10      # base.py:264, :283 -- no history, no convergence test before this


class LidDrivenCavitySolver(ABC):
    Parameters = None

    def __init__(self, params=None, **kwargs):
        if params is None:
            if self.Parameters is None:
                raise ValueError("Subclass must define Parameters class attribute")
            params = self.Parameters(**kwargs)      # unknown keys -> TypeError, like the reference
        self.params = params
        self.metrics = Metrics()
        self.fields = None
        self.time_series = None

    def _init_fields(self, x: np.ndarray, y: np.ndarray):
        n = len(x)
        self.fields = Fields(u=np.zeros(n), v=np.zeros(n), p=np.zeros(n), x=x.copy(), y=y.copy())

    # ---- what a device-backed subclass provides ------------------------------------------
    @abstractmethod
    def _begin(self, tolerance: float):
        """Prepare the device loop (tolerance latch, first dt)."""

    @abstractmethod
    def _advance(self, n_iters: int):
        """Run up to n_iters iterations; return (records[k, 8], latch, total_iterations)."""

    @abstractmethod
    def _finalize_fields(self):
        """Copy the final solution into self.fields."""

    @abstractmethod
    def compute_vortex_metrics(self) -> dict:
        ...

    # ---- driver ---------------------------------------------------------------------------
    def solve(self, tolerance: float = None, max_iter: int = None):
        tolerance = self.params.tolerance if tolerance is None else tolerance
        max_iter = self.params.max_iterations if max_iter is None else max_iter
        chunk = max(1, int(getattr(self.params, "check_every", 2048)))

        self._begin(tolerance)
        blocks, done, total = [], 0, 0
        t0 = time.perf_counter()
        while total < max_iter and not done:
            recs, done, total_new = self._advance(min(chunk, max_iter - total))
            if total_new == total:
                raise RuntimeError("device loop made no progress")
            blocks.append(recs)
            if log.isEnabledFor(logging.INFO) and len(recs):
                log.info("Iteration %d: rel=%.6e", total_new - 1, recs[-1, REL])
            total = total_new
        wall = time.perf_counter() - t0
        log.info("Solver finished in %.2f seconds.", wall)

        hist = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, 8))
        self._store_results(hist, total, done == 1, wall)

    @staticmethod
    def _downsample(values: list, limit: int):
        if values is None or len(values) <= limit:
            return values
        pick = np.linspace(0, len(values) - 1, limit, dtype=int)     # base.py:141
        return [values[i] for i in pick]

    def _store_results(self, hist: np.ndarray, iterations: int, converged: bool, wall: float,
                       max_timeseries_points: int = 1000):
        """hist holds one row per iteration (all of them); the reference keeps i >= 10 only."""
        kept = hist[WARMUP_ITERATIONS:]
        with_diag = bool(getattr(self.params, "diagnostics", True))
        col = lambda c: kept[:, c].tolist()                           # noqa: E731
        ds = lambda v: self._downsample(v, max_timeseries_points)     # noqa: E731
        self.history = hist
        self._finalize_fields()
        self.time_series = TimeSeries(
            rel_iter_residual=ds(col(REL)), u_residual=ds(col(RU)), v_residual=ds(col(RV)),
            continuity_residual=ds(col(RP)),
            energy=ds(col(EN)), enstrophy=ds(col(ZN) if with_diag else []),
            palinstrophy=ds(col(PN) if with_diag else []))
        try:
            vortex = self.compute_vortex_metrics()
        except Exception as exc:                                      # base.py:156-160
            log.warning("Failed to compute vortex metrics: %s", exc)
            vortex = {}
        last = kept[-1] if len(kept) else None
        self.metrics = Metrics(
            iterations=int(iterations), converged=bool(converged),
            final_residual=float(last[REL]) if last is not None else float("inf"),
            wall_time_seconds=float(wall),
            u_momentum_residual=float(last[RU]) if last is not None else 0.0,
            v_momentum_residual=float(last[RV]) if last is not None else 0.0,
            continuity_residual=float(last[RP]) if last is not None else 0.0,
            final_energy=float(last[EN]) if last is not None else 0.0,
            final_enstrophy=float(last[ZN]) if (last is not None and with_diag) else 0.0,
            final_palinstrophy=float(last[PN]) if (last is not None and with_diag) else 0.0,
            **{k: float(vortex.get(k, 0.0)) for k in _VORTEX_KEYS})
