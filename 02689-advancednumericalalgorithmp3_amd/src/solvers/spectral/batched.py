"""Batched trials: several independent SG solves of equal N advanced by the same launches.

The reference's only parallel axis is the trial (Hydra multirun / Optuna ``n_jobs`` / LSF arrays,
conf/machine/local.yaml:5-9, conf/experiment/optimization/corner_smoothing.yaml:53-57).  One N=128 trial
occupies 64 of the 256 CUs of an MI355X (one 16x16 tile per work-group), an N=64 trial 16, an N=32 trial 4;
batching B trials into every launch (``blockIdx.y`` = trial) fills the chip and amortises the per-launch
fixed cost.  Each trial keeps its own state, Re / lid profile / tolerance, dt, latch and history, so results
are bit-identical to running the trials one after another WITH THE SAME KERNEL (tests/test_gpu_batched.py).
Which kernel that is -- ``kernel_mode``: 0 the launch path, 3 one XCD per trial, 4 one CU per trial -- depends in
auto mode (``persistent=-1``) on the size and on how many trials share the batch (mode 4 from 80 trials, 32 at
N=32; a lone N>=80 trial takes mode 5, a batch of them the launch path).  The kernels agree to rounding
(<= 1e-12 over a fixture trajectory), not bit for bit, so an iteration count at the stopping threshold can move
by one with the batch size.  Every trial's ``results.json`` carries ``kernel_mode``; a study that must not depend
on batch size, world size or search-round size pins one (``LDC_PIN_MODE=0|3``, or ``search_mode=reference``
which pins 3: one XCD per trial where the size fits, the launch path above).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import logging
import time

import numpy as np

from . import ldc_lib as L
from .sg import SGSolver
from ..base import WARMUP_ITERATIONS

log = logging.getLogger(__name__)

LATCH_CAPPED = 3      # ctrl[DONE] code set by the host when a trial of a batch reaches its own max_iterations


class BatchedSGSolver:
    """``trials``: list of SGSolver keyword dicts with identical nx/ny (and device)."""
    kernel_mode = -1          # set when the batch handle exists (_ensure_batch)

    def __init__(self, trials: list):
        if not trials:
            raise ValueError("BatchedSGSolver needs at least one trial")
        if len({(int(t["nx"]), int(t["ny"])) for t in trials}) != 1:
            raise ValueError("all trials of a batch must share nx, ny")
        self.solvers = [SGSolver(**t) for t in trials]
        self._batch = None
        self._ws = None

    @classmethod
    def from_solvers(cls, solvers: list) -> "BatchedSGSolver":
        """Batch existing solver objects of equal N (e.g. the levels of several FSG trials)."""
        if not solvers or len({s.M for s in solvers}) != 1:
            raise ValueError("need at least one solver, all of the same N")
        self = cls.__new__(cls)
        self.solvers, self._batch, self._ws = list(solvers), None, None
        return self

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
        dev = self.solvers[0].device
        # The library fills the workspace by synchronous copies on a stream of its own, which it first makes wait for THIS stream
        # (ldc_batch_create's stream argument, ABI 7).  (The workspace used to be torch.zeros: a fill kernel queued on this
        # thread's stream.  Behind a long launch of another worker -- a chunk of the small-N or trial-per-CU kernel holds the chip
        # for tens of milliseconds -- the fill ran AFTER the library's copies and wiped the argument blocks: the batch's kernels
        # then read null pointers, "memory access fault on address (nil)" in a 600-trial search round.)
        self._ws = self._alloc_workspace(nbytes + 256, dev)
        base = (self._ws.data_ptr() + 255) & ~255
        arr = (C.c_void_p * B)(*[s._handle for s in self.solvers])
        h = C.c_void_p()
        with torch.cuda.device(dev):
            L.check(lib.ldc_batch_create(arr, B, base, nbytes, L.stream_ptr(dev), C.byref(h)), "ldc_batch_create")
        self._batch = h
        self._batch_keys = [s._handle_key for s in self.solvers]
        # which kernel advances these trials (0 launch path, 3 one XCD per trial, 4 one CU per trial): in auto mode that
        # depends on the size AND on how many trials share the batch -- every trial's record says which it was
        self.kernel_mode = int(lib.ldc_batch_mode(h))
        for s in self.solvers:
            s.kernel_mode = self.kernel_mode

    @staticmethod
    def _alloc_workspace(nbytes, dev):
        import torch
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)      # (nothing queued: the library overwrites what it needs)

    def close_batch(self):
        if self._batch is not None:
            import torch
            self.solvers[0]._sync()
            L.lib().ldc_batch_destroy(self._batch)
            self._batch = None

    def close(self):
        self.close_batch()
        for s in self.solvers:
            s.close()

    def run_to_tolerance(self, tolerances, max_iter, diagnostics: bool = False) -> list:
        """Every solver from its present state until ITS latch fires or ITS cap is reached (``max_iter``: one
        int for all, or one per solver); returns per-solver (latch, iterations, records).  A trial that hits its
        cap is latched on the device with code 3 (LATCH_CAPPED): its work-groups leave every later launch at
        entry, exactly like a converged trial's, and its history stops there -- the same outcome as the
        reference's one-process-per-trial runs with different ``max_iterations``.  Used by ``solve`` and by the
        batched FSG levels."""
        import torch
        n = len(self.solvers)
        caps = [int(max_iter)] * n if np.isscalar(max_iter) else [int(c) for c in max_iter]
        if len(caps) != n:
            raise ValueError("one iteration cap per solver expected")
        self._ensure_batch(list(tolerances))
        for s in self.solvers:
            s.d["ctrl"].zero_()
            s._prime()
        cap = min(s.rec_cap for s in self.solvers)
        chunk = max(1, min(min(int(s.params.check_every) for s in self.solvers), cap))
        blocks = [[] for _ in self.solvers]
        state = [(0, 0)] * n
        it = 0
        while True:
            live = [q for q in range(n) if not state[q][0]]
            for q in live:
                if it >= caps[q]:                                   # cap reached: latch it on the device
                    self.solvers[q].d["ctrl"][L.CTRL_DONE] = LATCH_CAPPED
                    state[q] = (LATCH_CAPPED, state[q][1])
            live = [q for q in live if not state[q][0]]
            if not live:
                break
            k = 1 if any(s._edge_fix_pending for s in self.solvers) else min(chunk, min(caps[q] for q in live) - it)
            for q, (rows, done, total) in enumerate(self._advance(k, diagnostics)):
                if state[q][0] != LATCH_CAPPED:
                    blocks[q].append(rows)
                    state[q] = (done, total)
            it += k
        self.solvers[0]._sync()
        return [(d, t, np.concatenate(b, axis=0) if b else np.zeros((0, 8))) for (d, t), b in zip(state, blocks)]

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
        # (one device-side gather and ONE copy per kind of word for the whole batch: a blocking copy per trial and kind --
        #  four of them -- was 12 ms of host time per chunk at 256 trials, as much as the chunk itself at N = 16)
        def words(key):
            return torch.stack([s.d[key] for s in self.solvers]).cpu().numpy()
        starts = [int(x) for x in words("ctrl")[:, L.CTRL_ITER]]
        resident = int(n_iters) > 1 and self.kernel_mode in (3, 4)
        lock = L.resident_lock(dev.index or 0) if resident else contextlib.nullcontext()
        with lock, torch.cuda.device(dev):        # (see ldc_lib.resident_lock: co-resident launches one at a time per device)
            L.check(L.lib().ldc_batch_enqueue(self._batch, int(n_iters), int(bool(diagnostics)), L.stream_ptr(dev)),
                    "ldc_batch_enqueue")
            self.solvers[0]._sync()
        out = []
        ctrl_all = words("ctrl")
        gave_up = torch.stack([s.d["sync"][L.SYNC_GIVEUP] for s in self.solvers]).cpu().numpy()
        same_cap = len({s.rec_cap for s in self.solvers}) == 1
        rings = words("rec") if same_cap else None
        for q, (s, start) in enumerate(zip(self.solvers, starts)):
            end, done = int(ctrl_all[q, L.CTRL_ITER]), int(ctrl_all[q, L.CTRL_DONE])
            ring = rings[q] if same_cap else s.d["rec"].cpu().numpy()
            out.append((ring[np.arange(start, end) % s.rec_cap], done, end))
            if int(gave_up[q]) != 0:
                raise L.LdcError("a persistent launch gave up a barrier wait (a work-group was not resident); the "
                                 "state of the batch is undefined -- rerun with persistent=0")
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
        """Every trial to its own tolerance or its own ``max_iterations``; fills each solver's metrics/fields.

        The trials share every launch, so only the batch has a wall time of its own (``self.batch_seconds``).
        A trial's ``metrics.wall_time_seconds`` is its SHARE of it, in proportion to its iteration count (the
        shares add up to the batch's wall time), so ``iterations / wall_time_seconds`` of any trial is the
        batch's aggregate rate in trial-iterations per second, not B times too low."""
        ps = [s.params for s in self.solvers]
        if len({bool(p.diagnostics) for p in ps}) != 1:
            raise ValueError("trials of one batch must agree on `diagnostics` (it selects the kernels of the "
                             "shared launches); batch them separately")
        caps = [p.max_iterations for p in ps] if max_iter is None else [max_iter] * len(ps)
        diag = bool(ps[0].diagnostics)
        t0 = time.perf_counter()
        out = self.run_to_tolerance([p.tolerance for p in ps], caps, diag)
        wall = time.perf_counter() - t0
        self.batch_seconds, self.batch_size = wall, len(self.solvers)
        its_all = max(1, sum(total for _, total, _ in out))
        for s, (done, total, hist) in zip(self.solvers, out):
            s.history = hist
            s._store_results(hist[WARMUP_ITERATIONS:], total, done == 1, wall * total / its_all)
        log.info("batched solve of %d trials finished in %.2f s", len(self.solvers), wall)
        return [s.metrics for s in self.solvers]


class BatchedFSGSolver:
    """Several FSG trials of equal N and hierarchy: level by level, the trials' level solvers share launches.

    Same sequence per trial as ``FSGSolver.solve`` (reference multigrid/fsg.py:1053-1221): every level runs in
    smoother mode to its own coarse tolerance with its own latch, a trial that diverges on a level stops there,
    the others go on; prolongation stays per trial (two small products).  Results are bit-identical to
    stand-alone FSG solves."""

    def __init__(self, trials: list):
        from .fsg import FSGSolver, hierarchy_orders
        if not trials:
            raise ValueError("BatchedFSGSolver needs at least one trial")
        self.solvers = [FSGSolver(**t) for t in trials]
        p0 = self.solvers[0].params
        self.orders = hierarchy_orders(p0.nx, p0.n_levels)
        for s in self.solvers:
            if hierarchy_orders(s.params.nx, s.params.n_levels) != self.orders:
                raise ValueError("all trials of a batch must share nx and the level hierarchy")

    def __len__(self):
        return len(self.solvers)

    def close(self):
        for s in self.solvers:
            s.close()

    def _run_level(self, group, tols, caps):
        """One level of all trials that are still alive.  A level the small kernels advance is ONE launch per chunk with every
        trial on an XCD / a CU of its own; a level on the launch path (N >= 80: the fine level of BASELINE config 5) gains what
        any launch-path batch gains from two halves on two streams (solve_concurrently: up to 1.3x) -- the coarse level of
        the same solve decides that the whole solve is one batch (main.py: run_batches), so the split happens here."""
        import os
        batch = BatchedSGSolver.from_solvers(group)
        n = len(group)
        split = n >= 4 and int(os.environ.get("LDC_BATCH_STREAMS", "3")) >= 2
        if split:
            batch._ensure_batch(list(tols))
            split = L.lib().ldc_batch_mode(batch._batch) == 0
        if not split:
            out = batch.run_to_tolerance(tols, caps, diagnostics=False)
            batch.close_batch()
            return out
        batch.close_batch()
        h = n // 2
        parts = [(group[:h], tols[:h], caps[:h]), (group[h:], tols[h:], caps[h:])]
        outs = [None, None]

        def run(k):
            b = BatchedSGSolver.from_solvers(parts[k][0])
            outs[k] = b.run_to_tolerance(parts[k][1], parts[k][2], diagnostics=False)
            b.close_batch()

        # streams of their own, not the pool's -- and not another pool worker's either: two FSG groups splitting a level at
        # the same time would otherwise take turns on one pair
        run_concurrently([0, 1], run, group[0].device, first_stream=8 + 2 * int(getattr(_WORKER, "index", 0)))
        return outs[0] + outs[1]

    def solve(self, max_iter: int = None):
        t0 = time.perf_counter()
        fines = self.solvers
        for s in fines:
            s._smoother_mode()
        ladders = [[s._make_level(n) for n in self.orders[:-1]] + [s] for s in fines]
        nlev = len(self.orders)
        alive = list(range(len(fines)))
        total = [0] * len(fines)
        last = [0] * len(fines)                       # latch of the last level each trial ran
        for idx in range(nlev):
            if not alive:          # every trial diverged on a coarser level (NaN latch): nothing goes up, as in FSGSolver.solve
                break
            group = [ladders[q][idx] for q in alive]
            for q, lvl in zip(alive, group):
                if idx == 0:
                    lvl.reset_state()
                else:
                    fines[q]._prolongate(ladders[q][idx - 1], lvl)
            tols = [fines[q].params.tolerance * fines[q].params.coarse_tolerance_factor ** (nlev - 1 - idx)
                    for q in alive]
            caps = [fines[q].params.max_iterations if max_iter is None else max_iter for q in alive]
            out = self._run_level(group, tols, caps)
            nxt = []
            for q, (done, its, _) in zip(alive, out):
                total[q] += its
                last[q] = done
                if done != 2:
                    nxt.append(q)
            log.info("batched FSG level %d (N=%d): %d trials, iterations %s", idx, self.orders[idx], len(alive),
                     [o[1] for o in out])
            alive = nxt
        wall = time.perf_counter() - t0
        self.batch_seconds, self.batch_size = wall, len(fines)
        its_all = max(1, sum(total))
        for q, s in enumerate(fines):
            for lvl in ladders[q][:-1]:
                lvl.close()
            # wall-time share by iteration count, see BatchedSGSolver.solve
            s._finish(s.params.tolerance, total[q], last[q] == 1 and q in alive, wall * total[q] / its_all)
        return [s.metrics for s in fines]


_WORKER_STREAMS = {}      # (device index, worker) -> torch.cuda.Stream, see run_concurrently
_WORKER_STREAMS_LOCK = __import__("threading").Lock()      # (several pool workers may ask for nested streams at once)
_WORKER = __import__("threading").local()                  # .index: which worker of run_concurrently this thread is (0 outside)


def run_concurrently(batches: list, fn, device=None, first_stream: int = 0) -> float:
    """``fn(batch)`` for every batch object AT THE SAME TIME, one host thread and one HIP stream each; returns the
    wall time of the lot.  The streams alternate between the two stream priorities HIP offers: streams of different
    priority never share a hardware queue, while which streams of ONE priority do is the runtime's choice (and on a
    shared queue nothing overlaps).  HIP has three priority levels on this hardware and torch hands out two of them, so
    the streams come from the library (``ldc_stream_create``) and are wrapped as ``torch.cuda.ExternalStream``: up to
    three workers with a hardware queue each (normal, high, low); a fourth and later ones repeat the cycle and may
    share.  An exception in a thread is re-raised here."""
    import threading

    import torch
    if len(batches) == 1:
        t0 = time.perf_counter()
        fn(batches[0])
        return time.perf_counter() - t0
    dev = torch.device(device if device is not None else "cuda")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    here = torch.cuda.current_stream(dev)
    # the worker streams are made once per device and kept: with a fresh pair per call the FOURTH pair of a process
    # ran its work three to five times slower, every time (config 5 at full size, round 4 of 8: 88 s instead of 20) --
    # which hardware queue a new stream lands on is the runtime's business, the first pair's placement is the measured one
    streams = []
    for k in range(len(batches)):
        key = (dev.index, first_stream + k)      # (first_stream: a nested use -- the levels of a batched FSG solve -- keeps off the pool's)
        with _WORKER_STREAMS_LOCK:
            if key not in _WORKER_STREAMS:
                with torch.cuda.device(dev):
                    least, greatest = C.c_int(), C.c_int()
                    L.check(L.lib().ldc_stream_priority_range(C.byref(least), C.byref(greatest)), "ldc_stream_priority_range")
                    levels = [0, greatest.value, least.value]               # normal, high, low
                    handle = C.c_void_p()
                    L.check(L.lib().ldc_stream_create(levels[k % 3], C.byref(handle)), "ldc_stream_create")
                _WORKER_STREAMS[key] = torch.cuda.ExternalStream(handle.value, device=dev)      # kept for the process' life
            streams.append(_WORKER_STREAMS[key])
    errors = [None] * len(batches)

    def work(k):
        try:
            _WORKER.index = first_stream + k
            with torch.cuda.device(dev), torch.cuda.stream(streams[k]):
                streams[k].wait_stream(here)          # the solvers were built (uploads, packing) on the caller's stream
                fn(batches[k])
                streams[k].synchronize()
        except BaseException as exc:                  # surfaces in the caller's thread below
            errors[k] = exc

    t0 = time.perf_counter()
    threads = [threading.Thread(target=work, args=(k,), name=f"ldc-batch-{k}") for k in range(len(batches))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    for exc in errors:
        if exc is not None:
            raise exc
    return wall


def solve_concurrently(batches: list, device=None) -> float:
    """``solve()`` of several batch objects (BatchedSGSolver / BatchedFSGSolver) at the same time (run_concurrently);
    returns the wall time of the lot.

    Why: a stage launch leaves the chip idle while it ramps up, drains and hands over to the next one (36 of the 51 us
    of an N=256 iteration are such fixed cost).  Launches of ANOTHER stream need not wait for that: as
    soon as a CU's work-group retires, one of the other batch moves in.  Two halves of a batch on two streams are
    never slower than the whole batch on one and up to 1.3x faster when the whole batch needs more than one round of
    work-groups (N=128, 8 trials: 114 k against 97 k trial-iterations/s; N=256, 2 trials: 23.4 k against 20.4 k;
    N=64, 32 trials: 483 k against 414 k -- tools/ab_streams.py).

    Every trial runs the same kernels in the same order as alone, so results stay bit-identical.  A trial's
    ``metrics.wall_time_seconds`` is rescaled so that the shares of all trials add up to the common wall time."""
    wall = run_concurrently(batches, lambda b: b.solve(), device)
    if len(batches) == 1:
        return batches[0].batch_seconds
    busy = sum(b.batch_seconds for b in batches)
    total = sum(len(b) for b in batches)
    for b in batches:
        for s in b.solvers:
            s.metrics.wall_time_seconds *= wall / busy
        b.batch_seconds, b.batch_size = wall, total
    log.info("%d batches (%d trials) on %d streams finished in %.2f s", len(batches), total, len(batches), wall)
    return wall
