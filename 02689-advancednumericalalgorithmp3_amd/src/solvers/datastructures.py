"""Boundary types of the solver plugin surface.

Field names, defaults and the ``to_mlflow`` conventions are the contract the reference's
launcher relies on (reference src/solvers/datastructures.py:29-51 Parameters, :59-109
Metrics, :117-143 TimeSeries, :151-165 Fields, :257-279 SpectralParameters); only the
names are shared, the implementation is this project's own.  YAML values override the
dataclass defaults (e.g. ``CFL: 1.5`` in conf/solver/spectral/sg.yaml).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field, fields
from typing import List, Optional

import numpy as np


def _mlflow_scalar(v):
    return int(v) if isinstance(v, bool) else v


class _Record:
    """Shared helpers: dict/DataFrame views of a dataclass."""

    def as_dict(self) -> dict:
        return {f.name: getattr(self, f.name) for f in fields(self)}

    def to_dataframe(self):
        import pandas as pd
        return pd.DataFrame([self.to_mlflow()])


@dataclass
class Parameters(_Record):
    name: str = ""
    Re: float = 100
    lid_velocity: float = 1.0
    Lx: float = 1.0
    Ly: float = 1.0
    nx: int = 64
    ny: int = 64
    max_iterations: int = 500
    tolerance: float = 1e-4
    method: str = ""

    def to_mlflow(self) -> dict:
        return {k: _mlflow_scalar(v) for k, v in self.as_dict().items()}


@dataclass
class SpectralParameters(Parameters):
    """nx/ny are the polynomial order N (N+1 collocation nodes per axis)."""
    basis_type: str = "legendre"
    CFL: float = 0.1
    beta_squared: float = 5.0
    method: str = "Spectral-AC"
    corner_treatment: str = "smoothing"
    corner_smoothing: float = 0.15
    multigrid: str = "none"
    n_levels: int = 3
    coarse_tolerance_factor: float = 10.0
    prolongation_method: str = "fft"
    restriction_method: str = "fft"
    # --- additions of the MI355X build (all optional, defaults keep reference behaviour) ---
    device: str = "cuda:0"
    check_every: int = 2048        # iterations enqueued between host polls of the latch
    graph_iters: int = 64          # iterations captured per hipGraph (32 -> 64: -0.2 us per N=256 iteration, flat beyond)
    nan_guard: bool = False        # quirk Q6: the reference SG spins on NaN; True = exit early
    diagnostics: bool = True       # E/Z/P every iteration, as base.py:274-276 does
    persistent: int = -1           # iteration loop: 0 = one launch per RK stage (hipGraph), 3 = the small-N kernel (one
                                   # launch per chunk, the trial on one XCD, N <= 79), 4 = the trial-per-CU kernel (M <= 44),
                                   # 5 = the chip-wide kernel (one launch per chunk, N = 81 ... 256), -1 = the library's
                                   # choice (include/ldc_hip.h): 3 / 5 where they apply; same results to rounding

    def to_mlflow(self) -> dict:
        skip = {"device", "check_every", "graph_iters", "nan_guard", "diagnostics", "persistent"}
        return {k: _mlflow_scalar(v) for k, v in self.as_dict().items() if k not in skip}


_VORTEX_KEYS = (
    "psi_min", "psi_min_x", "psi_min_y", "omega_center",
    "omega_max", "omega_max_x", "omega_max_y",
    "psi_BR", "omega_BR", "psi_BR_x", "psi_BR_y",
    "psi_BL", "omega_BL", "psi_BL_x", "psi_BL_y",
    "psi_TL", "omega_TL", "psi_TL_x", "psi_TL_y",
)


@dataclass
class Metrics(_Record):
    iterations: int = 0
    converged: bool = False
    final_residual: float = float("inf")
    wall_time_seconds: float = 0.0
    u_momentum_residual: float = 0.0
    v_momentum_residual: float = 0.0
    continuity_residual: float = 0.0
    final_energy: float = 0.0
    final_enstrophy: float = 0.0
    final_palinstrophy: float = 0.0
    psi_min: float = 0.0
    psi_min_x: float = 0.0
    psi_min_y: float = 0.0
    omega_center: float = 0.0
    omega_max: float = 0.0
    omega_max_x: float = 0.0
    omega_max_y: float = 0.0
    psi_BR: float = 0.0
    omega_BR: float = 0.0
    psi_BR_x: float = 0.0
    psi_BR_y: float = 0.0
    psi_BL: float = 0.0
    omega_BL: float = 0.0
    psi_BL_x: float = 0.0
    psi_BL_y: float = 0.0
    psi_TL: float = 0.0
    omega_TL: float = 0.0
    psi_TL_x: float = 0.0
    psi_TL_y: float = 0.0

    def to_mlflow(self) -> dict:
        """Bools as ints; values equal to +inf (unset) are dropped."""
        return {k: _mlflow_scalar(v) for k, v in self.as_dict().items() if v != math.inf}


@dataclass
class TimeSeries:
    rel_iter_residual: List[float] = field(default_factory=list)
    u_residual: List[float] = field(default_factory=list)
    v_residual: List[float] = field(default_factory=list)
    continuity_residual: Optional[List[float]] = field(default_factory=list)
    energy: List[float] = field(default_factory=list)
    enstrophy: List[float] = field(default_factory=list)
    palinstrophy: List[float] = field(default_factory=list)

    def items(self):
        return [(f.name, getattr(self, f.name)) for f in fields(self)]

    def to_records(self) -> list:
        """(key, step, value) triples; what the reference turns into mlflow Metric objects."""
        return [(k, step, v) for k, vals in self.items() if vals
                for step, v in enumerate(vals) if v is not None]

    def to_mlflow_batch(self) -> list:
        try:
            from mlflow.entities import Metric
        except ImportError:          # MLflow is optional in this build
            return self.to_records()
        return [Metric(key=k, value=v, timestamp=0, step=s) for k, s, v in self.to_records()]

    def to_dataframe(self):
        import pandas as pd
        return pd.DataFrame({k: v for k, v in self.items() if v})


@dataclass
class Fields:
    """Flat solution arrays of length (N+1)^2, C order of [ix, iy]."""
    u: np.ndarray
    v: np.ndarray
    p: np.ndarray
    x: np.ndarray
    y: np.ndarray

    def to_dataframe(self):
        import pandas as pd
        return pd.DataFrame({"x": self.x, "y": self.y, "u": self.u, "v": self.v, "p": self.p})
