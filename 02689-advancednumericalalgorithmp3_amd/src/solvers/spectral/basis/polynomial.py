"""Modal interpolation on collocation nodes -- the centreline-extraction recipe.

Contract: reference src/solvers/spectral/basis/polynomial.py:15-73 (Jacobi recurrence), :132-345
(derivative, Legendre-Gauss-Lobatto nodes / weights, Vandermonde matrices: the Legendre basis of the solver),
:398-477 (``spectral_interpolate``), used by the Ghia comparison
(src/shared/plotting/ldc/validation.py:297-322).  Post-processing, host side.
"""
from __future__ import annotations

import numpy as np


def jacobi_poly(xs, alpha: float, beta: float, N: int) -> np.ndarray:
    """P_N^{(alpha,beta)} at xs by the three-term recurrence."""
    xs = np.asarray(xs, dtype=float)
    older = np.ones_like(xs)
    if N == 0:
        return older
    newer = 0.5 * (alpha - beta + (alpha + beta + 2) * xs)
    ab = alpha + beta
    for m in range(1, N):
        s = 2 * m + ab
        lo = 2 * (m + alpha) * (m + beta) / ((s + 1) * s)
        mid = (alpha**2 - beta**2) / ((s + 2) * s) if alpha != beta else 0.0
        hi = 2 * (m + 1) * (m + ab + 1) / ((s + 2) * (s + 1))
        older, newer = newer, ((mid + xs) * newer - lo * older) / hi
    return newer


def grad_jacobi_poly(xs, alpha: float, beta: float, n: int):
    """d/dx P_n^{(alpha,beta)} = (alpha+beta+n+1)/2 P_{n-1}^{(alpha+1,beta+1)} (reference polynomial.py:132-157)."""
    if n == 0:
        return np.zeros_like(np.asarray(xs, dtype=float))
    return 0.5 * (alpha + beta + n + 1) * jacobi_poly(xs, alpha + 1, beta + 1, n - 1)


def legendre_gauss_lobatto_nodes(num_nodes: int) -> np.ndarray:
    """-1, the roots of P_N', +1 (reference polynomial.py:164-195)."""
    from numpy.polynomial.legendre import Legendre
    inner = Legendre.basis(num_nodes - 1).deriv().roots()
    return np.sort(np.concatenate(([-1.0], inner, [1.0])))


def legendre_gauss_lobatto_weights(num_nodes: int) -> np.ndarray:
    """2 / (N (N+1) P_N(x_j)^2) (reference polynomial.py:198-243)."""
    N = num_nodes - 1
    if N == 0:
        return np.array([2.0])
    return 2.0 / (N * (N + 1) * jacobi_poly(legendre_gauss_lobatto_nodes(num_nodes), 0.0, 0.0, N) ** 2)


def vandermonde_x(xs, alpha: float, beta: float) -> np.ndarray:
    return np.stack([grad_jacobi_poly(xs, alpha, beta, n) for n in range(len(xs))], axis=1)


def vandermonde(xs, alpha: float, beta: float, ncols: int | None = None) -> np.ndarray:
    """Columns P_0 ... P_{ncols-1} at xs from ONE pass of the recurrence: column n is what jacobi_poly(xs, alpha, beta, n)
    returns, operation for operation (a pass per column was 8 000 Python-level steps for a 129-node grid -- 0.13 s of
    the 0.17 s a trial's record took on the host, twice per record: the Ghia centreline error)."""
    ncols = len(xs) if ncols is None else ncols
    xs = np.asarray(xs, dtype=float)
    cols = np.empty((xs.size, ncols))
    if ncols > 0:
        cols[:, 0] = 1.0
    if ncols > 1:
        cols[:, 1] = 0.5 * (alpha - beta + (alpha + beta + 2) * xs)
    ab = alpha + beta
    for m in range(1, ncols - 1):
        s = 2 * m + ab
        lo = 2 * (m + alpha) * (m + beta) / ((s + 1) * s)
        mid = (alpha**2 - beta**2) / ((s + 2) * s) if alpha != beta else 0.0
        hi = 2 * (m + 1) * (m + ab + 1) / ((s + 2) * (s + 1))
        cols[:, m + 1] = ((mid + xs) * cols[:, m] - lo * cols[:, m - 1]) / hi
    return cols


def spectral_interpolate(x_nodes, f_values, x_eval, basis: str = "legendre") -> np.ndarray:
    """Evaluate the interpolating polynomial of (x_nodes, f_values) at x_eval."""
    table = {"legendre": (0.0, 0.0), "chebyshev": (-0.5, -0.5)}
    if basis.lower() not in table:
        raise ValueError(f"Unknown basis: {basis}. Use 'legendre' or 'chebyshev'.")
    alpha, beta = table[basis.lower()]
    x_nodes = np.asarray(x_nodes, dtype=float)
    x_eval = np.asarray(x_eval, dtype=float)
    lo, hi = x_nodes.min(), x_nodes.max()
    if not (np.isclose(lo, -1.0) and np.isclose(hi, 1.0)):
        x_nodes = 2.0 * (x_nodes - lo) / (hi - lo) - 1.0
        x_eval = 2.0 * (x_eval - lo) / (hi - lo) - 1.0
    modal = np.linalg.solve(vandermonde(x_nodes, alpha, beta), f_values)
    return vandermonde(x_eval, alpha, beta, ncols=len(x_nodes)) @ modal
