"""Chebyshev pseudospectral lid-driven-cavity solvers on MI355X."""
