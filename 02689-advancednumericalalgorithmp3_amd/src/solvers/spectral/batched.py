"""Batched trials: several independent SG solves of equal N advanced by the same launches.

The reference's only parallel axis is the trial (Hydra multirun / Optuna ``n_jobs`` / LSF arrays,
conf/machine/local.yaml:5-9, conf/experiment/optimization/corner_smoothing.yaml:53-57).  One N=128 trial
occupies 64 of the 256 CUs of an MI355X (one 16x16 tile per work-group), an N=64 trial 16, an N=32 trial 4;
batching B trials into every launch (``blockIdx.y`` = trial) fills the chip and amortises the per-launch
fixed cost.  Each trial keeps its own state, Re / lid profile / tolerance, dt, latch and history, so results
are bit-identical to running the trials one after another (tests/test_gpu_batched.py).
"""
from __future__ import annotations

import ctypes as C
import logging
import time

import numpy as np

from . import ldc_lib as L
from .sg import SGSolver
from ..base import WARMUP_ITERATIONS

log = logging.getLogger(__name__)


class BatchedSGSolver:
    """``trials``: list of SGSolver keyword dicts with identical nx/ny (and device)."""

    def __init__(self, trials: list):
        if not trials:
            raise ValueError("BatchedSGSolver needs at least one trial")
        if len({(int(t["nx"]), int(t["ny"])) for t in trials}) != 1:
            raise ValueError("all trials of a batch must share nx, ny")
        self.solvers = [SGSolver(**t) for t in trials]
        self._batch = None
        self._ws = None

    def __len__(self):
        return len(self.solvers)

    def _ensure_batch(self, tolerances):
        import torch
        for s, tol in zip(self.solvers, tolerances):
            s._ensure_handle(float(tol))
        if self._batch is not None and self._batch_keys == [s._handle_key for s in self.solvers]:
            return
        self.close_batch()
        lib = L.lib()
        B = len(self.solvers)
        nbytes = lib.ldc_batch_workspace_bytes(B)
        self._ws = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.solvers[0].device)
        base = (self._ws.data_ptr() + 255) & ~255
        arr = (C.c_void_p * B)(*[s._handle for s in self.solvers])
        h = C.c_void_p()
        L.check(lib.ldc_batch_create(arr, B, base, nbytes, C.byref(h)), "ldc_batch_create")
        self._batch = h
        self._batch_keys = [s._handle_key for s in self.solvers]

    def close_batch(self):
        if self._batch is not None:
            import torch
            torch.cuda.synchronize(self.solvers[0].device)
            L.lib().ldc_batch_destroy(self._batch)
            self._batch = None

    def close(self):
        self.close_batch()
        for s in self.solvers:
            s.close()

    def __del__(self):
        try:
            self.close_batch()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------
    def _advance(self, n_iters: int, diagnostics: bool):
        """Enqueue n_iters iterations for every trial; returns per-trial (rows, latch, total)."""
        import torch
        dev = self.solvers[0].device
        starts = [int(s.d["ctrl"].cpu().numpy()[L.CTRL_ITER]) for s in self.solvers]
        with torch.cuda.device(dev):
            L.check(L.lib().ldc_batch_enqueue(self._batch, int(n_iters), int(bool(diagnostics)), L.stream_ptr()),
                    "ldc_batch_enqueue")
            torch.cuda.synchronize(dev)
        out = []
        for s, start in zip(self.solvers, starts):
            ctrl = s.d["ctrl"].cpu().numpy()
            end, done = int(ctrl[L.CTRL_ITER]), int(ctrl[L.CTRL_DONE])
            ring = s.d["rec"].cpu().numpy()
            out.append((ring[np.arange(start, end) % s.rec_cap], done, end))
            if s._edge_fix_pending and end > start:
                s._write_boundary_edges(("U", "UT", "V", "VT"))
                s._edge_fix_pending = False
        return out

    def run_iterations(self, n: int, diagnostics: bool = True, tolerance: float = 0.0) -> list:
        """n more iterations for every trial (tolerance 0: no convergence stop); per-trial record arrays."""
        self._ensure_batch([tolerance] * len(self.solvers))
        for s in self.solvers:
            s._prime()
        rows = [[] for _ in self.solvers]
        left = int(n)
        cap = min(s.rec_cap for s in self.solvers)
        while left > 0:
            k = 1 if any(s._edge_fix_pending for s in self.solvers) else min(left, cap)
            for q, (r, _, _) in enumerate(self._advance(k, diagnostics)):
                rows[q].append(r)
            left -= k
        return [np.concatenate(r, axis=0) for r in rows]

    def solve(self, max_iter: int = None):
        """Every trial to its own tolerance (or max_iterations); fills each solver's metrics/fields."""
        p0 = self.solvers[0].params
        max_iter = p0.max_iterations if max_iter is None else max_iter
        diag = bool(p0.diagnostics)
        self._ensure_batch([s.params.tolerance for s in self.solvers])
        for s in self.solvers:
            s.d["ctrl"].zero_()
            s._prime()
        cap = min(s.rec_cap for s in self.solvers)
        chunk = max(1, min(int(p0.check_every), cap))
        blocks = [[] for _ in self.solvers]
        state = [(0, 0)] * len(self.solvers)            # (latch, total) per trial
        t0 = time.perf_counter()
        it = 0
        while it < max_iter and not all(d for d, _ in state):
            k = 1 if any(s._edge_fix_pending for s in self.solvers) else min(chunk, max_iter - it)
            res = self._advance(k, diag)
            for q, (rows, done, total) in enumerate(res):
                blocks[q].append(rows)
                state[q] = (done, total)
            it += k
        wall = time.perf_counter() - t0
        for q, s in enumerate(self.solvers):
            hist = np.concatenate(blocks[q], axis=0) if blocks[q] else np.zeros((0, 8))
            s.history = hist
            s._store_results(hist[WARMUP_ITERATIONS:], state[q][1], state[q][0] == 1, wall)
        log.info("batched solve of %d trials finished in %.2f s", len(self.solvers), wall)
        return [s.metrics for s in self.solvers]
