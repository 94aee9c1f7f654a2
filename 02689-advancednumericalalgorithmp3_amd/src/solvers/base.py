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

from . import validation as _val
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
            self._live_log(recs, total, bool(done == 1))
            total = total_new
        wall = time.perf_counter() - t0
        log.info("Solver finished in %.2f seconds.", wall)

        hist = np.concatenate(blocks, axis=0) if blocks else np.zeros((0, 8))
        self.history = hist
        # the reference keeps history only after the first 10 iterations (base.py:264)
        self._store_results(hist[WARMUP_ITERATIONS:], total, done == 1, wall)

    def _live_log(self, recs: np.ndarray, first: int, converged: bool, every: int = 50):
        """The reference's progress line and live MLflow metrics for every 50th iteration and the converged one
        (base.py:288-309).  The iterations of a chunk have already happened on the device when the host sees
        their records, so the same lines and metrics are emitted chunk by chunk; the metrics go out as ONE
        ``log_batch`` per chunk (same keys and ``step`` values as the reference's per-iteration
        ``log_metrics`` calls)."""
        n = len(recs)
        if n == 0:
            return
        idx = [k for k in range(n) if (first + k) % every == 0]
        if converged and (not idx or idx[-1] != n - 1):
            idx.append(n - 1)
        if not idx:
            return
        if log.isEnabledFor(logging.INFO):
            for k in idx:
                log.info("Iteration %d: rel=%.6e, u_res=%.6e, v_res=%.6e", first + k, recs[k, REL], recs[k, RU], recs[k, RV])
        try:
            import mlflow
        except ImportError:
            return
        run = mlflow.active_run()
        if run is None:
            return
        try:
            from mlflow.entities import Metric
            now = int(time.time() * 1000)
            with_diag = bool(getattr(self.params, "diagnostics", True))
            batch = []
            for k in idx:
                i = first + k
                live = {"rel_iter_residual": recs[k, REL], "u_residual": recs[k, RU], "v_residual": recs[k, RV],
                        "continuity_residual": recs[k, RP]}
                if i >= WARMUP_ITERATIONS:
                    live["energy"] = recs[k, EN]
                    if with_diag:
                        live["enstrophy"] = recs[k, ZN]
                batch.extend(Metric(key, float(val), now, i) for key, val in live.items())
            client = mlflow.tracking.MlflowClient()
            for lo in range(0, len(batch), 1000):            # MLflow's limit per log_batch call
                client.log_batch(run.info.run_id, metrics=batch[lo: lo + 1000])
        except Exception as exc:                              # tracking must never stop a solve
            log.warning("live metrics not logged: %s", exc)

    @staticmethod
    def _downsample(values: list, limit: int):
        if values is None or len(values) <= limit:
            return values
        pick = np.linspace(0, len(values) - 1, limit, dtype=int)     # base.py:141
        return [values[i] for i in pick]

    def _store_results(self, kept: np.ndarray, iterations: int, converged: bool, wall: float,
                       max_timeseries_points: int = 1000, with_diag: bool = None):
        """`kept`: one row per recorded iteration (columns REL..DT)."""
        if with_diag is None:
            with_diag = bool(getattr(self.params, "diagnostics", True))
        col = lambda c: kept[:, c].tolist()                           # noqa: E731
        ds = lambda v: self._downsample(v, max_timeseries_points)     # noqa: E731
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

    # ---- validation against stored FV solutions (reference base.py:970-1054, 1122-1160) -----
    def _evaluate_at_points(self, x: np.ndarray, y: np.ndarray):
        """Bilinear evaluation of (u, v) on the tensor grid of ``self.fields``; NaN outside."""
        from scipy.interpolate import RegularGridInterpolator
        xs, ys = np.sort(np.unique(self.fields.x)), np.sort(np.unique(self.fields.y))
        order = np.lexsort((self.fields.x, self.fields.y))          # y slow, x fast
        pts = np.column_stack([y, x])
        out = []
        for f in (self.fields.u, self.fields.v):
            grid = f[order].reshape(ys.size, xs.size)
            out.append(RegularGridInterpolator((ys, xs), grid, method="linear", bounds_error=False,
                                               fill_value=np.nan)(pts))
        return out[0], out[1]

    @staticmethod
    def _load_reference_solution(directory):
        """(x, y, u, v) of a stored FV solution: the reference's ``solution.vts`` or the compact npz."""
        from pathlib import Path
        d = Path(directory)
        if (d / "solution.vts").exists():
            from .vtkio import read_vts
            g = read_vts(d / "solution.vts")
            return g["points"][:, 0], g["points"][:, 1], g["point_data"]["u"], g["point_data"]["v"]
        if (d / "solution.npz").exists():
            g = np.load(d / "solution.npz")
            return g["x"], g["y"], g["u"], g["v"]
        return None

    def compute_validation_errors(self, reference_dir: str = "data/validation/fv", save_plots: bool = False) -> dict:
        """Relative L2 errors of u, v against the FV solutions at their 128 x 128 cell centres.

        Like the reference (quirk Q9) the argument is ignored: both ``fv`` and ``fv-regu`` under
        ``data/validation`` are compared (CWD first, then the packaged copies)."""
        res = {}
        Re = int(self.params.Re)
        base = _val.find_data_dir()
        for sub, suffix in (("fv", ""), ("fv-regu", "_regu")):
            ref = self._load_reference_solution(base / sub / f"Re{Re}")
            if ref is None:
                continue
            rx, ry, ru, rv = ref
            cu, cv = self._evaluate_at_points(rx, ry)
            eps = 1e-10
            ok = ((rx > eps) & (rx < self.params.Lx - eps) & (ry > eps) & (ry < self.params.Ly - eps)
                  & ~(np.isnan(cu) | np.isnan(cv)))
            res[f"u_L2_error{suffix}"] = float(np.linalg.norm(cu[ok] - ru[ok]) / (np.linalg.norm(ru[ok]) + 1e-12))
            res[f"v_L2_error{suffix}"] = float(np.linalg.norm(cv[ok] - rv[ok]) / (np.linalg.norm(rv[ok]) + 1e-12))
        return res

    def ghia_error(self) -> dict:
        """Centreline error against Ghia et al. 1982 (SURVEY.md 8d Metric 2)."""
        shape = getattr(self, "shape_full", None)          # (nodes along x, nodes along y); nx != ny allowed
        if shape is None:
            M = int(round(np.sqrt(self.fields.x.size)))
            shape = (M, M)
        x, y = self.fields.x.reshape(shape)[:, 0], self.fields.y.reshape(shape)[0, :]
        return _val.ghia_centerline_error(x, y, self.fields.u.reshape(shape), self.fields.v.reshape(shape),
                                          int(self.params.Re))

    def validation_table(self) -> list:
        return _val.botella_table(self.metrics, int(self.params.Re))

    def mlflow_log_validation_table(self, reference_csv: str = None):
        """Rows are always returned; they go to MLflow only when it is importable and a run is active."""
        rows = self.validation_table()
        try:
            import mlflow
            import pandas as pd
            if rows and mlflow.active_run():
                mlflow.log_table(pd.DataFrame(rows), artifact_file="validation_metrics.json")
        except ImportError:
            pass
        return rows

    # ---- VTS export (reference base.py:464-549) ---------------------------------------------------
    def to_vtk(self):
        from scipy.interpolate import RectBivariateSpline
        from .vtkio import StructuredGridFile
        xs, ys = np.sort(np.unique(self.fields.x)), np.sort(np.unique(self.fields.y))
        order = np.lexsort((self.fields.x, self.fields.y))
        U, V, P = (f[order].reshape(ys.size, xs.size) for f in (self.fields.u, self.fields.v, self.fields.p))
        g = StructuredGridFile(xs, ys)
        g["u"], g["v"], g["pressure"] = U.ravel(), V.ravel(), P.ravel()
        g["velocity_magnitude"] = np.sqrt(U**2 + V**2).ravel()
        # Quirk Q11 (reference base.py:545-549), reproduced on purpose: the splines are built over
        # (y, x) but the reference asks for dx=1 / dy=1 meaning "d/dx" / "d/dy"; scipy differentiates
        # along the FIRST / SECOND argument, so the exported array is dv/dy - du/dx, not the vorticity.
        # The solver's own vorticity (metrics, psi) uses the spectral operators and is unaffected.
        first = RectBivariateSpline(ys, xs, V)(ys, xs, dx=1)
        second = RectBivariateSpline(ys, xs, U)(ys, xs, dy=1)
        g["vorticity"] = (first - second).ravel()
        g["velocity"] = np.column_stack([U.ravel(), V.ravel(), np.zeros(U.size)])
        g.field_data.update(Re=np.array([self.params.Re]), N=np.array([self.params.nx]),
                            solver=np.array([self.params.name]))
        return g

    def save_vtk(self, filepath):
        self.to_vtk().save(filepath)
