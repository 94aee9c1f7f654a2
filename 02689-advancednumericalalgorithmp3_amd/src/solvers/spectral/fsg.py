"""Full-single-grid (FSG) sequence on MI355X: converge on N/2 (and coarser), prolong, converge on N.

Plugin surface: ``solvers.spectral.fsg.FSGSolver`` as in conf/solver/spectral/fsg.yaml:7 (reference
src/solvers/spectral/fsg.py:19-129; driver ``solve_fsg`` multigrid/fsg.py:1053-1221, hierarchy :489-543,
prolongation :551-614, smoother :745-995).  Every level is the same fused RK-stage kernel as SGSolver,
run in *smoother* mode: each stage differentiates its own stage pressure (the reference smoother passes
``p_in`` through, unlike SGSolver -- quirk Q1), no warm-up, NaN/Inf exit.  Prolongation = two NT products
with a precomputed matrix on the GPU; the axis-swapped boundary re-imposition of the reference (quirk Q2)
is reproduced because it changes the first fine-level stage.
"""
from __future__ import annotations

import dataclasses
import logging
import time

import numpy as np

from . import ldc_lib as L
from .operators.transfer_operators import prolongation_matrix
from .sg import SGSolver
from ..base import EN, PN, REL, RP, RU, RV, ZN

log = logging.getLogger(__name__)

COARSEST_N = 12           # multigrid/fsg.py:496


def hierarchy_orders(n_fine: int, n_levels: int, coarsest_n: int = COARSEST_N) -> list:
    """Polynomial orders coarse -> fine: halve while the next order stays >= coarsest_n."""
    orders, n = [], n_fine
    for _ in range(n_levels):
        orders.append(n)
        if n // 2 < coarsest_n:
            break
        n //= 2
    return orders[::-1]


class FSGSolver(SGSolver):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if self.params.nx != self.params.ny:
            # the reference builds its level hierarchy from ONE order (multigrid/fsg.py:490-530, n_fine = nx) and squares
            # every level; with nx != ny it has no defined behaviour to reproduce
            raise NotImplementedError("FSG needs nx == ny (the level hierarchy is built from one polynomial order)")

    def _smoother_mode(self):
        self._stage_pressure, self._warmup, self._nan_exit = 1, 0, True

    def _make_level(self, n: int) -> SGSolver:
        kw = dataclasses.asdict(dataclasses.replace(self.params, nx=n, ny=n))
        lvl = SGSolver(**kw)
        lvl._stage_pressure, lvl._warmup, lvl._nan_exit = 1, 0, True
        return lvl

    # ------------------------------------------------------------------ prolongation (device)
    def _prolongate(self, coarse: SGSolver, fine: SGSolver):
        import torch
        LD, Mf, Mc = fine.LD, fine.M, coarse.M
        R = (Mf + 15) // 16
        dev = fine.device
        method = self.params.prolongation_method

        def padded(a):
            out = torch.zeros((LD, LD), dtype=torch.float64, device=dev)
            out[: a.shape[0], : a.shape[1]] = torch.as_tensor(a, device=dev) if not torch.is_tensor(a) else a
            return out

        Pf = padded(prolongation_matrix(method, Mc, Mf))                 # (Mf, Mc)
        Pi = padded(prolongation_matrix(method, Mc - 2, Mf - 2))         # inner grids
        X, Y = fine.d["S0"], fine.d["S1"]

        def apply(P, src_T, n_out):
            """P src P^T for a coarse field given as its transpose (LD-padded); returns LD x LD."""
            # X[i][b] = sum_a P[i][a] src[a][b] = sum_a P[i][a] srcT[b][a]
            fine._abi("ldc_gemm_nt", P.data_ptr(), src_T.data_ptr(), X.data_ptr(), R, R, LD, 0, 0, None, None)
            # out[i][j] = sum_b X[i][b] P[j][b]
            fine._abi("ldc_gemm_nt", X.data_ptr(), P.data_ptr(), Y.data_ptr(), R, R, LD, 0, 0, None, None)
            return Y[:n_out, :n_out].clone()

        uc = padded(coarse.d["UT"][:Mc, :Mc])
        u = apply(Pf, uc, Mf)
        vc = padded(coarse.d["VT"][:Mc, :Mc])
        v = apply(Pf, vc, Mf)
        pc = padded(coarse.d["P"][1: Mc - 1, 1: Mc - 1].t().contiguous())
        p = apply(Pi, pc, Mf - 2)
        # quirk Q2 (multigrid/fsg.py:586-599): [ix, iy] arrays treated as [iy, ix]
        U0 = float(self.params.lid_velocity)
        u[0, :] = 0.0; v[0, :] = 0.0
        u[-1, :] = U0; v[-1, :] = 0.0
        u[:, 0] = 0.0; v[:, 0] = 0.0
        u[:, -1] = 0.0; v[:, -1] = 0.0
        # initialize_lid (:950-954) restores the lid column with the regularised profile
        u[:, -1] = fine.d["ulid"][:Mf]
        v[:, -1] = 0.0
        fine.set_state_device(u, v, p)

    def _finish(self, tolerance, total, converged, wall):
        """One-point histories, like the reference (fsg.py:102-124)."""
        res = self.residual_fields()
        q = self.global_quantities()
        row = np.zeros((1, 8))
        row[0, REL] = tolerance if converged else tolerance * 10
        row[0, RU], row[0, RV], row[0, RP] = (float(np.linalg.norm(res[k])) for k in ("R_u", "R_v", "R_p"))
        row[0, EN], row[0, ZN], row[0, PN] = q["E"], q["Z"], q["P"]
        self.history = row
        self._store_results(row, total, converged, wall, with_diag=True)

    # ------------------------------------------------------------------ driver
    def solve(self, tolerance: float = None, max_iter: int = None):
        tolerance = self.params.tolerance if tolerance is None else tolerance
        max_iter = self.params.max_iterations if max_iter is None else max_iter
        p = self.params
        t0 = time.perf_counter()
        orders = hierarchy_orders(p.nx, p.n_levels)
        log.info("Building %d-level hierarchy: N = %s", len(orders), orders)
        self._smoother_mode()
        levels = [self._make_level(n) for n in orders[:-1]] + [self]
        chunk = max(1, int(p.check_every))
        total, converged, diverged = 0, False, False
        for idx, lvl in enumerate(levels):
            tol = tolerance * p.coarse_tolerance_factor ** (len(levels) - 1 - idx)
            if idx == 0:
                lvl.reset_state()
            else:
                self._prolongate(levels[idx - 1], lvl)
            keep = lvl.params.diagnostics
            lvl.params.diagnostics = False
            try:
                lvl._begin(float(tol))
                done, it = 0, 0
                while it < max_iter and not done:
                    _, done, it = lvl._advance(min(chunk, max_iter - it))
            finally:
                lvl.params.diagnostics = keep
            total += it
            converged = done == 1
            log.info("FSG level %d (N=%d): %d iterations, converged=%s", idx, lvl.params.nx, it, converged)
            if done == 2:
                diverged = True
                break
        for lvl in levels[:-1]:
            lvl.close()
        wall = time.perf_counter() - t0
        converged = bool(converged and not diverged)
        self._finish(tolerance, total, converged, wall)
        log.info("FSG completed in %.2fs: %d iterations, converged=%s", wall, total, converged)
