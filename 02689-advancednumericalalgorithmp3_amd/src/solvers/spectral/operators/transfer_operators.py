"""Coarse-to-fine transfer for the FSG sequence, as precomputed matrices.

Contract: reference ``FFTProlongation`` (src/solvers/spectral/operators/transfer_operators.py:209-255)
applied row/column-wise (:93-129).  The operator is linear, so it is built ONCE per level pair as an
(n_fine x n_coarse) matrix on the host and applied on the GPU as two NT products.  Two reference
behaviours are kept on purpose: the end samples are weighted twice (halved by hand
and again inside the un-normalised DCT-I), so the operator is not an interpolation; and the inner-grid
pressure is pushed through the same CGL-based operator.  Only prolongation is reached by FSG; the
restriction / injection operators of the reference are dead code for this path and are not provided.
"""
from __future__ import annotations

import numpy as np


def fft_prolongation_matrix(n_coarse: int, n_fine: int) -> np.ndarray:
    """Matrix of the DCT-based prolongation between grids of n_coarse and n_fine points."""
    if n_coarse == n_fine:
        return np.eye(n_coarse)
    if n_coarse > n_fine:
        raise ValueError(f"Prolongation requires n_coarse ({n_coarse}) <= n_fine ({n_fine})")
    order_c, order_f = n_coarse - 1, n_fine - 1
    modes = np.arange(n_coarse)
    # un-normalised DCT-I as a matrix: y_k = x_0 + (-1)^k x_N + 2 sum_{0<n<N} x_n cos(pi k n / N)
    transform = 2.0 * np.cos(np.pi * np.outer(modes, modes) / order_c)
    transform[:, 0] = 1.0
    transform[:, -1] = (-1.0) ** modes
    halve = np.ones(n_coarse)
    halve[[0, -1]] = 0.5
    coefficients = halve[:, None] * transform * halve[None, :] / order_c
    evaluate = np.cos(np.pi * np.outer(np.arange(n_fine), modes) / order_f)
    return evaluate @ coefficients


def polynomial_prolongation_matrix(n_coarse: int, n_fine: int) -> np.ndarray:
    """Exact Chebyshev interpolation between CGL grids (the reference's 'polynomial' option)."""
    from numpy.polynomial.chebyshev import chebvander
    xc = -np.cos(np.pi * np.arange(n_coarse) / (n_coarse - 1))
    xf = -np.cos(np.pi * np.arange(n_fine) / (n_fine - 1))
    return chebvander(xf, n_coarse - 1) @ np.linalg.solve(chebvander(xc, n_coarse - 1), np.eye(n_coarse))


def prolongation_matrix(method: str, n_coarse: int, n_fine: int) -> np.ndarray:
    key = method.lower()
    if key == "fft":
        return fft_prolongation_matrix(n_coarse, n_fine)
    if key == "polynomial":
        return polynomial_prolongation_matrix(n_coarse, n_fine)
    raise ValueError(f"Unknown prolongation method: {method}. Use 'fft' or 'polynomial'.")
