"""Lid-driven-cavity solvers -- MI355X-native spectral path.

Only the spectral solvers are provided (the finite-volume solver of the reference is out
of scope); ``_target_: solvers.spectral.sg.SGSolver`` resolves here unchanged.
"""
from .datastructures import Fields, Metrics, Parameters, SpectralParameters, TimeSeries  # noqa: F401
