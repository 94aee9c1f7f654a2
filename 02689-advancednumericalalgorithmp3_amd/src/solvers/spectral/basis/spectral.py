"""Chebyshev-Gauss-Lobatto collocation basis (host-side setup, NumPy).

Behavioural contract: reference src/solvers/spectral/basis/spectral.py
(:18-39 nodes, :42-90 differentiation matrix, :411-470 Clenshaw-Curtis, :473-541 basis
class).  Setup is O(N^2) host work done once per solver; the per-step arithmetic lives in
csrc/ldc_kernels.hip.  Only the Chebyshev basis is provided: the Legendre and Fourier
bases of the reference are outside the hot path (conf/solver/spectral/sg.yaml:19).
"""
from __future__ import annotations

import numpy as np


def chebyshev_gauss_lobatto_nodes(num_points: int) -> np.ndarray:
    """Ascending extrema of T_N on [-1, 1]: xi_j = -cos(pi j / N)."""
    n = num_points - 1
    return -np.cos(np.pi * np.arange(num_points) / n)


def chebyshev_diff_matrix(nodes: np.ndarray) -> np.ndarray:
    """Collocation derivative on the given CGL nodes (reference interval).

    Off-diagonal entries follow the classical (c_i/c_j)(-1)^(i+j)/(x_i-x_j) formula; each
    diagonal entry is minus the sum of its row so constants differentiate to exactly 0
    (and to match the reference bit-for-bit the row sum is numpy's pairwise ``np.sum``).
    """
    m = nodes.size
    if m == 1:
        return np.zeros((1, 1))
    c = np.ones(m)
    c[[0, -1]] = 2.0
    k = np.arange(m)
    gap = nodes[:, None] - nodes[None, :]
    gap[k, k] = 1.0
    d = (c[:, None] / c[None, :]) * (-1.0) ** (k[:, None] + k[None, :]) / gap
    d[k, k] = 0.0
    for row in range(m):
        d[row, row] = -np.sum(d[row, :])
    return d


def clenshaw_curtis_weights(num_points: int) -> np.ndarray:
    """Quadrature weights on the CGL nodes of [-1, 1] (they sum to 2)."""
    n = num_points - 1
    if n == 0:
        return np.array([2.0])
    if n == 1:
        return np.array([1.0, 1.0])
    modes = np.arange(n // 2 + 1)
    coeff = np.where(modes == 0, 1.0, 2.0 / (1.0 - 4.0 * modes * modes))
    if n % 2 == 0:
        coeff[-1] *= 0.5
    j = np.arange(num_points)
    w = (2.0 / n) * (np.cos(2.0 * np.pi * np.outer(j, modes) / n) @ coeff)
    w[[0, -1]] *= 0.5
    return w


class ChebyshevLobattoBasis:
    """CGL nodes / derivative / weights affinely mapped to ``domain``."""

    def __init__(self, domain=(-1.0, 1.0)):
        self.domain = (float(domain[0]), float(domain[1]))

    @property
    def _half_length(self) -> float:
        return 0.5 * (self.domain[1] - self.domain[0])

    def nodes(self, num_points: int) -> np.ndarray:
        xi = chebyshev_gauss_lobatto_nodes(num_points)
        if self.domain == (-1.0, 1.0):
            return xi
        a, b = self.domain
        return 0.5 * (b - a) * (xi + 1.0) + a

    def diff_matrix(self, nodes: np.ndarray) -> np.ndarray:
        a, b = self.domain
        return (2.0 / (b - a)) * chebyshev_diff_matrix(chebyshev_gauss_lobatto_nodes(nodes.size))

    def quadrature_weights(self, num_points: int) -> np.ndarray:
        a, b = self.domain
        return clenshaw_curtis_weights(num_points) * (b - a) / 2


def legendre_diff_matrix(nodes: np.ndarray) -> np.ndarray:
    """D = Vx V^-1 on arbitrary nodes, Legendre modes (reference basis/spectral.py:93-130)."""
    from .polynomial import vandermonde, vandermonde_x
    return vandermonde_x(nodes, 0.0, 0.0) @ np.linalg.solve(vandermonde(nodes, 0.0, 0.0), np.eye(nodes.size))


class LegendreLobattoBasis:
    """LGL nodes / derivative / weights affinely mapped to ``domain`` (reference basis/spectral.py:326-407)."""

    def __init__(self, domain=(-1.0, 1.0)):
        self.domain = (float(domain[0]), float(domain[1]))

    def nodes(self, num_points: int) -> np.ndarray:
        from .polynomial import legendre_gauss_lobatto_nodes
        xi = legendre_gauss_lobatto_nodes(num_points)
        if self.domain == (-1.0, 1.0):
            return xi
        a, b = self.domain
        return 0.5 * (b - a) * (xi + 1.0) + a

    def diff_matrix(self, nodes: np.ndarray) -> np.ndarray:
        from .polynomial import legendre_gauss_lobatto_nodes
        a, b = self.domain
        return (2.0 / (b - a)) * legendre_diff_matrix(legendre_gauss_lobatto_nodes(nodes.size))

    def quadrature_weights(self, num_points: int) -> np.ndarray:
        from .polynomial import legendre_gauss_lobatto_weights
        a, b = self.domain
        return legendre_gauss_lobatto_weights(num_points) * (b - a) / 2


def inner_to_full_interpolation(nodes_inner: np.ndarray, nodes_full: np.ndarray) -> np.ndarray:
    """(M, M-2) matrix evaluating, on all nodes, the polynomial of degree M-3 that
    interpolates values given on the interior nodes (P_N-P_{N-2} pressure; reference
    sg.py:212-248 -- Chebyshev Vandermonde quotient)."""
    from numpy.polynomial.chebyshev import chebvander
    a, b = nodes_full[0], nodes_full[-1]
    to_ref = lambda x: 2 * (x - a) / (b - a) - 1  # noqa: E731
    n = nodes_inner.size
    v_inner = chebvander(to_ref(nodes_inner), n - 1)
    v_full = chebvander(to_ref(nodes_full), n - 1)
    return v_full @ np.linalg.solve(v_inner, np.eye(n))
