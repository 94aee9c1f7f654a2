"""Lid-velocity regularisation at the two top corners.

Contract: reference src/solvers/spectral/operators/corner.py (:64-123 cosine smoothing of
width ``corner_smoothing * Lx``, :131-180 the 16 s^2 (1-s)^2 profile, :192-223 factory and
its error message).  The profile is computed ONCE here and uploaded; the kernels apply it
as the north boundary value of every RK stage.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


class CornerTreatment:
    def get_lid_velocity(self, x, y, lid_velocity: float, Lx: float, Ly: float) -> Tuple[np.ndarray, np.ndarray]:
        raise NotImplementedError

    def get_wall_velocity(self, x, y, Lx: float, Ly: float) -> Tuple[np.ndarray, np.ndarray]:
        shape = np.asarray(x).shape
        return np.zeros(shape), np.zeros(shape)

    def uses_modified_convection(self) -> bool:
        return False


class SmoothingTreatment(CornerTreatment):
    """u = U * (1 - cos(pi d / delta)) / 2 within distance delta of a corner, U elsewhere."""

    def __init__(self, smoothing_width: float = 0.15):
        self.smoothing_width = smoothing_width

    def get_lid_velocity(self, x, y, lid_velocity, Lx, Ly):
        x = np.asarray(x, dtype=float)
        u = np.full(x.shape, float(lid_velocity))
        if self.smoothing_width > 0:
            delta = self.smoothing_width * Lx
            ramp = lambda d: 0.5 * (1 - np.cos(np.pi * d / delta)) * lid_velocity  # noqa: E731
            near_left = x < delta                  # strict, like the reference masks
            near_right = x > (Lx - delta)
            u[near_left] = ramp(x[near_left])
            u[near_right] = ramp(Lx - x[near_right])
        return u, np.zeros(x.shape)


class SaadTreatment(CornerTreatment):
    """u = 16 s^2 (1-s)^2 U with s = x / Lx."""

    def get_lid_velocity(self, x, y, lid_velocity, Lx, Ly):
        s = np.asarray(x, dtype=float) / Lx
        return 16.0 * s**2 * (1.0 - s) ** 2 * lid_velocity, np.zeros(s.shape)


PolynomialTreatment = SaadTreatment


def create_corner_treatment(method: str = "smoothing", smoothing_width: float = 0.15, **_) -> CornerTreatment:
    key = method.lower()
    if key == "smoothing":
        return SmoothingTreatment(smoothing_width=smoothing_width)
    if key in ("polynomial", "saad"):
        return SaadTreatment()
    raise ValueError(
        f"Unknown corner treatment method: {method}. Use 'smoothing', 'polynomial', or 'saad'.")
