"""Single-grid Chebyshev P_N-P_{N-2} artificial-compressibility solver on MI355X.

Plugin surface: ``solvers.spectral.sg.SGSolver(**cfg.solver)`` exactly as the reference's
Hydra ``_target_`` (conf/solver/spectral/sg.yaml:3; reference class
src/solvers/spectral/sg.py:29).  Same constructor keys, same ``solve()/metrics/fields/
time_series``; the arithmetic of ``step()`` (sg.py:410-449), the per-iteration norms and
E/Z/P (base.py:250-276) and the psi post-processing (sg.py:556-709) run in hand-written
gfx950 kernels behind ``libldc_hip.so``; this file only builds operators with NumPy, owns
the device tensors and polls a latch.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import logging

import numpy as np

from ..base import LidDrivenCavitySolver
from ..datastructures import SpectralParameters
from . import ldc_lib as L
from .basis.spectral import ChebyshevLobattoBasis, LegendreLobattoBasis, inner_to_full_interpolation
from .operators.corner import create_corner_treatment

log = logging.getLogger(__name__)

_MAT_NAMES = ("Dx", "D2x", "Dy", "D2y", "IxF", "GxF", "IyF", "GyF",
              "U", "UT", "V", "VT", "P", "UA", "UAT", "VA", "VAT", "PA",
              "UB", "UBT", "VB", "VBT", "PB", "T1T", "T2T", "PX", "PY", "W", "WT",
              "S0", "S1", "S2", "S3", "S4", "S5", "S6", "S7", "S8", "S9", "S10",   # S*: scratch
              # packed twins (include/ldc_hip.h, Conventions): what the MFMA operand loads of the loop read
              "DxK", "D2xK", "DyK", "D2yK", "IxFK", "GxFK", "IyFK", "GyFK",
              "UK", "UTK", "VK", "VTK", "PK", "UAK", "UATK", "VAK", "VATK", "PAK",
              "UBK", "UBTK", "VBK", "VBTK", "PBK", "T1TK", "T2TK", "WK", "WTK")
_PACKED_OPERATORS = ("Dx", "D2x", "Dy", "D2y", "IxF", "GxF", "IyF", "GyF")
_PACKED_STATE = ("U", "UT", "V", "VT", "P", "UA", "UAT", "VA", "VAT", "PA", "UB", "UBT", "VB", "VBT", "PB")
_DEBUG_KEYS = ("du_dx", "du_dy", "dv_dx", "dv_dy", "lap_u", "lap_v", "dp_dx", "dp_dy", "R_u", "R_v", "R_p")


_EIG_CACHE = {}
_AXIS_CACHE = {}


def _axis_operators(basis, kind: str, M: int):
    """(nodes, D, D^2, inner-to-full interpolation, quadrature weights) of one axis.  Kept per (basis, domain, M) and handed out
    read-only: the trials of a sweep differ in Re and in the lid, not in their grids, and building these was ~50 ms per solver."""
    key = (str(kind).lower(), tuple(float(v) for v in basis.domain), int(M))
    hit = _AXIS_CACHE.get(key)
    if hit is None:
        x = basis.nodes(M)
        D = basis.diff_matrix(x)
        hit = (x, D, D @ D, inner_to_full_interpolation(x[1:-1], x), basis.quadrature_weights(M))
        for a in hit:
            a.setflags(write=False)
        if len(_AXIS_CACHE) > 64:
            _AXIS_CACHE.clear()
        _AXIS_CACHE[key] = hit
    return hit


def _interior_eigenbasis(D2: np.ndarray):
    """(lam, Q, Q^-1) of the interior block of a second-derivative matrix.  Kept per matrix (its bytes): the trials of a sweep
    share their operators, and the two decompositions were ~0.1 s of host time per trial at N=128."""
    import hashlib
    blk = np.ascontiguousarray(D2[1:-1, 1:-1])
    key = (blk.shape, hashlib.sha1(blk.tobytes()).hexdigest())
    hit = _EIG_CACHE.get(key)          # (worker threads of a sweep share the cache: read once, never re-read after a clear)
    if hit is None:
        lam, Q = np.linalg.eig(blk)
        if np.max(np.abs(lam.imag)) > 1e-8 * np.max(np.abs(lam.real)):
            raise RuntimeError("interior D2 block has complex eigenvalues")
        lam, Q = lam.real, Q.real
        hit = (lam, Q, np.linalg.inv(Q))
        if len(_EIG_CACHE) > 32:
            _EIG_CACHE.clear()
        _EIG_CACHE[key] = hit
    return hit


class _ArraysView:
    """``solver.arrays.u / .v / .p`` as host copies of the device state (flat, like the
    reference's SpectralSolverFields); assigning uploads."""

    def __init__(self, owner):
        object.__setattr__(self, "_o", owner)

    def __getattr__(self, name):
        o = self._o
        if name in ("u", "v"):
            return o._download_full("U" if name == "u" else "V").ravel()
        if name == "p":
            return o._download_full("P")[1:-1, 1:-1].ravel().copy()
        if name in ("R_u", "R_v", "R_p") or name in _DEBUG_KEYS:
            return o.residual_fields()[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        o = self._o
        if name not in ("u", "v", "p"):
            raise AttributeError(name)
        cur = {"u": None, "v": None, "p": None}
        cur[name] = np.asarray(value, dtype=float)
        o.set_state(**cur)


class SGSolver(LidDrivenCavitySolver):
    Parameters = SpectralParameters

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        p = self.params
        kind = p.basis_type.lower()
        if kind == "chebyshev":
            self.basis_x = ChebyshevLobattoBasis(domain=(0.0, p.Lx))
            self.basis_y = ChebyshevLobattoBasis(domain=(0.0, p.Ly))
        elif kind == "legendre":          # same device path: only the host-built operators differ
            self.basis_x = LegendreLobattoBasis(domain=(0.0, p.Lx))
            self.basis_y = LegendreLobattoBasis(domain=(0.0, p.Ly))
        else:
            raise ValueError(f"Unknown basis_type: {p.basis_type}. Use 'legendre' or 'chebyshev'")
        self.corner_treatment = create_corner_treatment(
            method=p.corner_treatment, smoothing_width=p.corner_smoothing)

        self._build_operators()
        self.shape_full = (self.Mx, self.My)
        self.shape_inner = (self.Mx - 2, self.My - 2)
        self._init_fields(x=self.x_full.ravel(), y=self.y_full.ravel())
        self._alloc_device()
        self.arrays = _ArraysView(self)
        self._handle = None
        self._handle_tol = None
        self._eig = None
        # loop flavour: SG (reference sg.py/base.py) unless a subclass switches to the FSG smoother
        self._stage_pressure = 0          # 1: every RK stage differentiates its own stage pressure
        self._warmup = 10                 # iterations without convergence test (base.py:264, 283)
        self._nan_exit = None             # None: follow params.nan_guard
        self._edge_fix_pending = False
        self.reset_state()

    # ------------------------------------------------------------------ host-side setup
    def _build_operators(self):
        p = self.params
        # Independent x / y grids (reference sg.py:103-119).  nx != ny: the device arrays, the tiling and LD are built for
        # M = max(Mx, My); operators, vectors and fields of the shorter axis are zero padded like everything beyond M, the
        # kernels take the node classes (wall, lid, interior) from (Mx, My), and every path that takes such grids (launch per
        # stage, one-XCD kernel, chip-wide kernel) runs in the layout that keeps index M-1 inside the tiles
        # (include/ldc_hip.h, ldc_problem::Mx).
        Mx, My = p.nx + 1, p.ny + 1
        M = max(Mx, My)
        self.M, self.Mx, self.My = M, Mx, My
        x, self.Dx_1d, self.Dxx_1d, self.Interp_x, self.w_x = _axis_operators(self.basis_x, p.basis_type, Mx)
        y, self.Dy_1d, self.Dyy_1d, self.Interp_y, self.w_y = _axis_operators(self.basis_y, p.basis_type, My)
        self.x_nodes, self.y_nodes = x, y
        self.x_full, self.y_full = np.meshgrid(x, y, indexing="ij")
        self.dx_min = float(np.min(np.diff(x)))
        self.dy_min = float(np.min(np.diff(y)))
        u_lid, _ = self.corner_treatment.get_lid_velocity(
            x, np.full_like(x, p.Ly), lid_velocity=p.lid_velocity, Lx=p.Lx, Ly=p.Ly)
        self.u_lid = u_lid
        # geometry of the MFMA tiling
        # (N a multiple of 16 up to 256: index M-1 stays outside the tiles so that T*T work-groups fill
        #  the chip exactly; beyond that the grid is large anyway and index M-1 gets a tile row of its own)
        self.T = (M - 1 + 15) // 16
        self.tail = 1 if (16 * self.T == M - 1 and self.T <= 16 and Mx == My) else 0
        if 16 * self.T == M - 1 and not self.tail:
            self.T += 1
        self.LD = 16 * self.T + 16

    def _alloc_device(self):
        import torch
        dev = torch.device(self.params.device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        L.require_device(dev)
        self.device = dev
        LD, M = self.LD, self.M
        self._mats = torch.zeros((len(_MAT_NAMES), LD, LD), dtype=torch.float64, device=dev)
        self.d = {n: self._mats[k] for k, n in enumerate(_MAT_NAMES)}
        self._vecs = torch.zeros((12, LD), dtype=torch.float64, device=dev)
        for k, n in enumerate(("wx", "wy", "ulid", "x", "y", "lamx", "lamy", "DxL", "D2xL", "DyL", "D2yL")):
            self.d[n] = self._vecs[k]
        nt = self.T * self.T
        n_edge = (2 * M - 1 + 3) // 4 if self.tail else 0
        self._part_stride = (nt + n_edge) * L.NPART
        self.d["partials"] = torch.zeros(5 * self._part_stride, dtype=torch.float64, device=dev)
        self.d["scal"] = torch.zeros(L.SCAL_LEN, dtype=torch.float64, device=dev)
        self.d["ctrl"] = torch.zeros(L.CTRL_LEN, dtype=torch.int32, device=dev)
        # barrier words of the persistent trial kernel (256-byte aligned: torch's allocator hands out 512-byte blocks)
        self.d["sync"] = torch.zeros(L.SYNC_LEN, dtype=torch.int32, device=dev)
        self.rec_cap = max(1, int(self.params.check_every))
        self.d["rec"] = torch.zeros((self.rec_cap, L.REC_LEN), dtype=torch.float64, device=dev)
        self.d["ext_val"] = torch.zeros(8, dtype=torch.float64, device=dev)
        self.d["ext_idx"] = torch.zeros(8, dtype=torch.int32, device=dev)

        def up(name, a):
            pad = np.zeros((LD, LD))
            pad[: a.shape[0], : a.shape[1]] = a
            self.d[name].copy_(torch.from_numpy(pad))

        IxF = np.zeros((self.Mx, self.Mx)); IxF[:, 1:-1] = self.Interp_x
        IyF = np.zeros((self.My, self.My)); IyF[:, 1:-1] = self.Interp_y
        # The interpolant through the inner nodes evaluated AT an inner node is that node's value: rows 1 .. M-2 are unit
        # rows up to the rounding of V_full V_inner^-1 (1e-15).  They are set exactly (include/ldc_hip.h: the small-N trial
        # kernel contracts GxF with p itself off the ring); anything but rounding there would be a different operator.
        for F in (IxF, IyF):
            inner = F[1:-1, 1:-1]
            if np.max(np.abs(inner - np.eye(inner.shape[0]))) > 1e-9:
                raise ValueError("inner-to-full interpolation is not the identity on the inner nodes")
            inner[...] = np.eye(inner.shape[0])
        for name, a in (("Dx", self.Dx_1d), ("D2x", self.Dxx_1d), ("Dy", self.Dy_1d), ("D2y", self.Dyy_1d),
                        ("IxF", IxF), ("GxF", self.Dx_1d @ IxF), ("IyF", IyF), ("GyF", self.Dy_1d @ IyF)):
            up(name, a)
        self._pack(_PACKED_OPERATORS)
        for name, v in (("wx", self.w_x), ("wy", self.w_y), ("ulid", self.u_lid),
                        ("x", self.x_nodes), ("y", self.y_nodes),
                        ("DxL", self.Dx_1d[:, -1]), ("D2xL", self.Dxx_1d[:, -1]),
                        ("DyL", self.Dy_1d[:, -1]), ("D2yL", self.Dyy_1d[:, -1])):
            pad = np.zeros(LD); pad[: len(v)] = v
            self.d[name].copy_(torch.from_numpy(pad))

    def _abi(self, fn: str, *args):
        """One C-ABI launch with THIS solver's device current and on its current stream (always the
        last argument): the library launches on, and sets per-device kernel attributes for, the
        current HIP device, whatever device the caller happens to have selected."""
        import torch
        with torch.cuda.device(self.device):
            L.check(getattr(L.lib(), fn)(*args, L.stream_ptr(self.device)), fn)

    def _pack(self, names):
        """Bring the packed twins of the named row-major arrays up to date (host-side edits only:
        the kernels keep both forms in step themselves)."""
        for n in names:
            self._abi("ldc_pack", self.d[n].data_ptr(), self.d[n + "K"].data_ptr(), self.LD)

    # ------------------------------------------------------------------ state transfer
    def _download_full(self, name: str) -> np.ndarray:
        import torch
        self._sync()
        return self.d[name][: self.Mx, : self.My].cpu().numpy()

    def _upload_full(self, name: str, a2d: np.ndarray, transposed_name: str = None):
        import torch
        pad = np.zeros((self.LD, self.LD))
        pad[: a2d.shape[0], : a2d.shape[1]] = a2d
        t = torch.from_numpy(pad).to(self.device)
        self.d[name].copy_(t)
        if transposed_name:
            self.d[transposed_name].copy_(t.t())

    def set_state(self, u=None, v=None, p=None):
        """Upload (flat or 2-D) host arrays; p lives on the (N-1)^2 inner grid."""
        Mx, My = self.Mx, self.My
        if u is not None:
            self._upload_full("U", np.asarray(u, float).reshape(Mx, My), "UT")
        if v is not None:
            self._upload_full("V", np.asarray(v, float).reshape(Mx, My), "VT")
        if p is not None:
            full = np.zeros((Mx, My))
            full[1:-1, 1:-1] = np.asarray(p, float).reshape(Mx - 2, My - 2)
            self._upload_full("P", full)
        for src, dsts in (("U", ("UA", "UB")), ("UT", ("UAT", "UBT")), ("V", ("VA", "VB")),
                          ("VT", ("VAT", "VBT")), ("P", ("PA", "PB"))):
            for dn in dsts:
                self.d[dn].copy_(self.d[src])
        if self.tail:
            # Index M-1 is never rewritten by the tiles.  The RK stage buffers must carry the
            # boundary VALUES there (the reference re-imposes them after every stage); phi^n itself
            # keeps whatever was uploaded for the first iteration and is corrected right after it.
            self._write_boundary_edges(("UA", "UAT", "VA", "VAT"))
            self._write_boundary_edges(("UB", "UBT", "VB", "VBT"))
            self._edge_fix_pending = True
        self._pack(_PACKED_STATE)
        self._primed = False

    def _write_boundary_edges(self, names):
        """Boundary values on the row/column of index M-1 of (U, UT, V, VT)-like arrays (tail layout only:
        index M-1 = 16 T lies in block row/column T, which no MFMA operand load reads, so the packed
        twins need no update)."""
        m1, M = self.M - 1, self.M
        U, UT, V, VT = (self.d[n] for n in names)
        lid = self.d["ulid"][:M]
        U[m1, :M] = 0.0; U[:M, m1] = lid          # east wall, then the lid (lid wins the corner)
        UT[:M, m1] = 0.0; UT[m1, :M] = lid
        V[m1, :M] = 0.0; V[:M, m1] = 0.0
        VT[:M, m1] = 0.0; VT[m1, :M] = 0.0

    def set_state_device(self, u, v, p_inner):
        """Same as set_state for torch tensors already on the device ((M,M), (M,M), (M-2,M-2))."""
        Mx, My = self.Mx, self.My
        for name, tname, a in (("U", "UT", u), ("V", "VT", v)):
            self.d[name].zero_(); self.d[tname].zero_()
            self.d[name][:Mx, :My] = a
            self.d[tname][:My, :Mx] = a.t()
        self.d["P"].zero_()
        self.d["P"][1: Mx - 1, 1: My - 1] = p_inner
        self.set_state()

    def reset_state(self):
        """Fluid at rest with the regularised lid (reference sg.py:76-98)."""
        Mx, My = self.Mx, self.My
        u = np.zeros((Mx, My))
        u[:, -1] = self.u_lid
        self.set_state(u=u, v=np.zeros((Mx, My)), p=np.zeros((Mx - 2, My - 2)))
        self.d["ctrl"].zero_()
        self.d["scal"].zero_()

    # ------------------------------------------------------------------ C-ABI plumbing
    def _problem(self, tol: float) -> L.Problem:
        p = self.params
        pr = L.Problem()
        pr.M, pr.LD, pr.T, pr.tail = self.M, self.LD, self.T, self.tail
        pr.Mx, pr.My = self.Mx, self.My
        pr.nu, pr.beta2, pr.cfl = 1.0 / p.Re, p.beta_squared, p.CFL
        pr.hx_min, pr.hy_min, pr.lid_speed, pr.tol = self.dx_min, self.dy_min, p.lid_velocity, tol
        nan_exit = bool(p.nan_guard) if self._nan_exit is None else bool(self._nan_exit)
        pr.warmup, pr.nan_guard, pr.stage_pressure, pr.rec_cap = (
            int(self._warmup), int(nan_exit), int(self._stage_pressure), self.rec_cap)
        for name, _ in L.Problem._fields_:
            if name in self.d and name != "partials_stride":
                setattr(pr, name, self.d[name].data_ptr())
        pr.partials_stride = self._part_stride
        return pr

    def _ensure_handle(self, tol: float):
        key = (tol, self._stage_pressure, self._warmup, self._nan_exit, int(self.params.persistent))
        if self._handle is not None and self._handle_key == key:
            return
        self._handle_key = key
        self.close()
        h = C.c_void_p()
        pr = self._problem(tol)
        import torch
        with torch.cuda.device(self.device):      # the handle belongs to the device that is current here
            L.check(L.lib().ldc_solver_create(C.byref(pr), C.byref(h)), "ldc_solver_create")
        # owned from here on: a setter that fails below must not leak the handle (close() destroys it)
        self._handle, self._handle_tol = h, tol
        try:
            L.check(L.lib().ldc_solver_set_graph_iters(h, int(self.params.graph_iters)), "ldc_solver_set_graph_iters")
            mode = int(self.params.persistent)
            # A persistent mode on a size it cannot run (mode 3: more tiles than one XCD holds; mode 4: more state than one
            # CU's LDS holds; mode 5: fewer than 6 x 6 or more than 16 x 16 tiles, e.g. the levels of a hierarchy) falls back
            # to the launch path -- all the same way:
            # a configuration asks for "persistent where it applies", the levels of one FSG solve differ in size.  The
            # library knows the device (CUs, XCDs) and says LDC_E_ARG; nothing is asked of torch here: this runs in the
            # worker threads of a sweep, where torch.cuda.get_device_properties once raised "Invalid device id" (round 3).
            # (The silent process aborts of that round at test_gpu_config5_shape_batched_fsg_vs_oracle have no proven cause --
            #  DESIGN.md 3 says what was changed and what is inferred; this is one of the two changes kept as hygiene.)
            rc = L.lib().ldc_solver_set_persistent(h, mode)
            if rc == -1 and mode in (3, 4, 5):
                rc = L.lib().ldc_solver_set_persistent(h, 0)
            L.check(rc, "ldc_solver_set_persistent")
            self.kernel_mode = int(L.lib().ldc_solver_mode(h))       # (a batch overwrites it with ITS mode: batched.py)
        except Exception:
            self.close()
            self._handle_key = None
            raise

    def _sync(self):
        """Wait for this solver's work: everything it enqueues goes to the calling thread's current stream of its
        device (``_abi``).  NOT a device-wide synchronise: with two batches advanced from two host threads
        (batched.run_concurrently) that would also wait for the other thread's stream, and HIP refuses it outright
        while the other thread captures a graph."""
        import torch
        torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        if getattr(self, "_handle", None) is not None:
            import torch
            self._sync()
            L.lib().ldc_solver_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _prime(self):
        if not self._primed:
            self._abi("ldc_prime", self._handle)
            self._primed = True

    # ------------------------------------------------------------------ driver hooks
    def _begin(self, tolerance: float, restart: bool = True):
        """Every solve() counts iterations from 0 (warm-up included), like base.py:243."""
        self._ensure_handle(float(tolerance))
        if restart:
            self.d["ctrl"].zero_()
        self._prime()

    def _advance(self, n_iters: int):
        import torch
        ctrl0 = self.d["ctrl"].cpu().numpy()
        start = int(ctrl0[L.CTRL_ITER])
        n_iters = min(int(n_iters), self.rec_cap)
        if self._edge_fix_pending and n_iters > 1:
            # first iteration after an upload: afterwards phi^n carries its boundary values
            rows1, done1, end1 = self._advance(1)
            if done1 or n_iters == 1:
                return rows1, done1, end1
            rows2, done2, end2 = self._advance(n_iters - 1)
            return np.concatenate([rows1, rows2], axis=0), done2, end2
        if n_iters > 1 and L.lib().ldc_solver_mode(self._handle) != 0:
            # a launch whose work-groups must be co-resident: one at a time per device (ldc_lib.resident_lock)
            with L.resident_lock(self.device.index or 0):
                self._abi("ldc_solver_enqueue", self._handle, n_iters, int(bool(self.params.diagnostics)))
                self._sync()
        else:
            self._abi("ldc_solver_enqueue", self._handle, n_iters, int(bool(self.params.diagnostics)))
            self._sync()
        ctrl = self.d["ctrl"].cpu().numpy()
        end, done = int(ctrl[L.CTRL_ITER]), int(ctrl[L.CTRL_DONE])
        if int(self.d["sync"][L.SYNC_GIVEUP]) != 0:
            raise L.LdcError("persistent trial kernel gave up a barrier wait (a work-group was not resident); "
                             "the state is undefined -- rerun with persistent=0")
        ring = self.d["rec"].cpu().numpy()
        rows = ring[np.arange(start, end) % self.rec_cap]
        if self._edge_fix_pending and end > start:
            self._write_boundary_edges(("U", "UT", "V", "VT"))
            self._edge_fix_pending = False
        return rows, done, end

    def step(self):
        """One pseudo-time step (4 RK stages + BCs) and the reductions that give the next dt."""
        self._begin(self.params.tolerance if self._handle_tol is None else self._handle_tol, restart=False)
        self._advance_raw(1, diagnostics=False)
        return self.arrays.u, self.arrays.v, self.arrays.p

    def _advance_raw(self, n, diagnostics):
        keep = self.params.diagnostics
        self.params.diagnostics = diagnostics
        try:
            return self._advance(n)
        finally:
            self.params.diagnostics = keep

    def run_iterations(self, n: int, diagnostics: bool = True, tolerance: float = 0.0,
                       restart: bool = False) -> np.ndarray:
        """Run n more iterations (tolerance 0 = no convergence stop); returns their records."""
        self._begin(tolerance, restart=restart)
        out, left = [], int(n)
        while left > 0:
            rows, done, _ = self._advance_raw(min(left, self.rec_cap), diagnostics)
            out.append(rows)
            left -= len(rows)
            if done or len(rows) == 0:
                break
        return np.concatenate(out, axis=0) if out else np.zeros((0, 8))

    # ------------------------------------------------------------------ results
    def _finalize_fields(self):
        u, v, pf = self._download_full("U"), self._download_full("V"), self._download_full("P")
        self.fields.u[:] = u.ravel()
        self.fields.v[:] = v.ravel()
        self.fields.p[:] = self._extrapolate_to_full_grid(pf[1:-1, 1:-1]).ravel()

    def _extrapolate_to_full_grid(self, inner: np.ndarray) -> np.ndarray:
        """Output pressure: linear extrapolation to the edges (quirk Q5; reference sg.py:144-179)."""
        f = np.zeros(self.shape_full)
        f[1:-1, 1:-1] = inner
        f[0, 1:-1] = 2 * f[1, 1:-1] - f[2, 1:-1]
        f[-1, 1:-1] = 2 * f[-2, 1:-1] - f[-3, 1:-1]
        f[1:-1, 0] = 2 * f[1:-1, 1] - f[1:-1, 2]
        f[1:-1, -1] = 2 * f[1:-1, -2] - f[1:-1, -3]
        for (a, b), (n1, n2) in ((((0, 0)), ((0, 1), (1, 0))), ((0, -1), ((0, -2), (1, -1))),
                                 ((-1, 0), ((-1, 1), (-2, 0))), ((-1, -1), ((-1, -2), (-2, -1)))):
            f[a, b] = 0.5 * (f[n1] + f[n2])
        return f

    def residual_fields(self, which: int = 0) -> dict:
        """All intermediates of one residual evaluation (parity tests; reference sg.py:278-346)."""
        import torch
        self._ensure_handle(self.params.tolerance if self._handle_tol is None else self._handle_tol)
        outs = [self.d[f"S{k}"] for k in range(11)]
        for t in outs:
            t.zero_()
        arr = (C.c_void_p * 11)(*[t.data_ptr() for t in outs])
        self._abi("ldc_residual_debug", self._handle, which, arr)
        self._sync()
        res = {}
        for k, key in enumerate(_DEBUG_KEYS):
            a = outs[k][: self.Mx, : self.My].cpu().numpy()
            res[key] = a[1:-1, 1:-1].ravel().copy() if key == "R_p" else a.ravel()
        return res

    def global_quantities(self) -> dict:
        """E, Z, P of the current state, computed on the device (reference sg.py:495-550)."""
        import torch
        self._ensure_handle(self.params.tolerance if self._handle_tol is None else self._handle_tol)
        out = self.d["ext_val"]
        self._abi("ldc_global_quantities", self._handle, out.data_ptr())
        self._sync()
        e, z, p = out[:3].cpu().numpy()
        return {"E": float(e), "Z": float(z), "P": float(p)}

    # ---- vorticity / stream function / vortices (reference sg.py:510-743) --------------------
    def _compute_vorticity(self) -> np.ndarray:
        import torch
        self._ensure_handle(self.params.tolerance if self._handle_tol is None else self._handle_tol)
        self._abi("ldc_diagnostics", self._handle)
        self._sync()
        return self._download_full("W").ravel()

    def _eigenbasis(self):
        """Eigen-decomposition of the interior second-derivative blocks (done once, host)."""
        if self._eig is None:
            import torch
            out = {}
            for tag, D2 in (("x", self.Dxx_1d), ("y", self.Dyy_1d)):
                out[tag] = _interior_eigenbasis(D2)
            names = {"Qx": out["x"][1], "Qxi": out["x"][2], "Qy": out["y"][1], "Qyi": out["y"][2]}
            self._eig_t = torch.zeros((4, self.LD, self.LD), dtype=torch.float64, device=self.device)
            for k, (n, a) in enumerate(names.items()):
                pad = np.zeros((self.LD, self.LD)); pad[: a.shape[0], : a.shape[1]] = a
                self._eig_t[k].copy_(torch.from_numpy(pad))
            for n, lam in (("lamx", out["x"][0]), ("lamy", out["y"][0])):
                pad = np.full(self.LD, -1.0); pad[: lam.size] = lam      # padding: any non-zero sum
                self.d[n].copy_(torch.from_numpy(pad))
            self._eig = out
        return self._eig_t

    def _compute_streamfunction(self):
        """psi from lap(psi) = -omega, psi = 0 on the walls, by fast diagonalisation on the GPU."""
        import torch
        Q = self._eigenbasis()
        self._compute_vorticity()
        Mx, My, Mi = self.Mx, self.My, self.M - 2          # (Mi: the larger inner block; the eigenbases are zero padded)
        F, w0, w1, Psi = self.d["S0"], self.d["S1"], self.d["S2"], self.d["S3"]
        F.zero_()
        F[: Mx - 2, : My - 2] = -self.d["W"][1: Mx - 1, 1: My - 1]
        self._abi("ldc_poisson_fastdiag",
                  Q[0].data_ptr(), Q[1].data_ptr(), Q[2].data_ptr(), Q[3].data_ptr(),
                  self.d["lamx"].data_ptr(), self.d["lamy"].data_ptr(), F.data_ptr(), w0.data_ptr(),
                  w1.data_ptr(), Psi.data_ptr(), Mi, self.LD)
        full = self.d["S4"]
        full.zero_()
        full[1: Mx - 1, 1: My - 1] = Psi[: Mx - 2, : My - 2]
        self._sync()
        return full[:Mx, :My].cpu().numpy(), self.x_full, self.y_full

    def compute_vortex_metrics(self) -> dict:
        import torch
        self._compute_streamfunction()          # leaves psi in S4 and omega in W
        self._abi("ldc_vortex_extrema_xy",
                  self.d["S4"].data_ptr(), self.d["W"].data_ptr(), self.d["x"].data_ptr(), self.d["y"].data_ptr(),
                  self.Mx, self.My, self.LD, self.d["ext_val"].data_ptr(), self.d["ext_idx"].data_ptr())
        self._sync()
        val = self.d["ext_val"].cpu().numpy()
        idx = self.d["ext_idx"].cpu().numpy()
        W = self.d["W"].cpu().numpy()
        at = lambda k: divmod(int(idx[k]), self.LD)          # noqa: E731
        x, y = self.x_nodes, self.y_nodes
        i, j = at(0)
        out = dict(psi_min=float(val[0]), psi_min_x=float(x[i]), psi_min_y=float(y[j]),
                   omega_center=float(W[i, j]))
        i, j = at(1)
        out.update(omega_max=float(val[1]), omega_max_x=float(x[i]), omega_max_y=float(y[j]))
        for k, name in ((2, "BR"), (3, "BL"), (4, "TL")):
            i, j = at(k)
            if val[k] > 0:
                vals = (float(val[k]), float(W[i, j]), float(x[i]), float(y[j]))
            else:
                vals = (0.0, 0.0, 0.0, 0.0)
            out[f"psi_{name}"], out[f"omega_{name}"], out[f"psi_{name}_x"], out[f"psi_{name}_y"] = vals
        return out
