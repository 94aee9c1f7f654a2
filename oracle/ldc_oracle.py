"""CPU oracle for the Chebyshev P_N-P_{N-2} artificial-compressibility lid-driven-cavity path.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker / reported CPU baseline.  The product path (``solvers.spectral.sg.SGSolver`` ->
``libldc_hip.so``) never routes through it.

It is a plain NumPy restatement of the reference algorithm, written from the mathematics
(SURVEY.md section 8a "algorithm card"), each function citing the reference lines whose
behaviour it reproduces (paths relative to ``/root/reference``).  Parity is PINNED: the
golden vectors in ``tests/golden/*.npz`` were produced by importing and running the
reference itself in the build container (``tests/golden/make_golden.py``), and
``tests/test_oracle_golden.py`` checks every function below against them.

Array convention (``src/solvers/spectral/sg.py:108``): 2-D fields are ``F[ix, iy]``,
C-order, y fastest; ``D @ F`` differentiates in x, ``F @ D.T`` in y.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
from numpy.polynomial.chebyshev import chebvander

RK_ALPHAS = (0.25, 1.0 / 3.0, 0.5, 1.0)  # sg.py:430


# --------------------------------------------------------------------------- 1-D operators
def cgl_nodes(M: int) -> np.ndarray:
    """xi_j = -cos(pi j / N), j = 0..N, ascending in [-1, 1]  (basis/spectral.py:18-39)."""
    N = M - 1
    return -np.cos(np.pi * np.arange(M) / N)


def cheb_diff_reference_interval(xi: np.ndarray) -> np.ndarray:
    """Chebyshev collocation derivative on [-1, 1] (basis/spectral.py:42-90).

    Off-diagonal (c_i/c_j)(-1)^(i+j)/(xi_i - xi_j) with c_0 = c_N = 2; the diagonal is
    the negative row sum, taken with ``np.sum`` row by row like the reference (quirk Q7).
    """
    M = xi.size
    if M == 1:
        return np.zeros((1, 1))
    c = np.ones(M)
    c[0] = c[-1] = 2.0
    idx = np.arange(M)
    sign = (-1.0) ** (idx[:, None] + idx[None, :])
    dx = xi[:, None] - xi[None, :]
    np.fill_diagonal(dx, 1.0)
    D = (c[:, None] / c[None, :]) * sign / dx
    np.fill_diagonal(D, 0.0)
    for i in range(M):
        D[i, i] = -np.sum(D[i, :])
    return D


def clenshaw_curtis(M: int) -> np.ndarray:
    """Clenshaw-Curtis weights on the CGL nodes of [-1, 1], sum = 2 (basis/spectral.py:411-470)."""
    N = M - 1
    if N == 0:
        return np.array([2.0])
    if N == 1:
        return np.array([1.0, 1.0])
    k = np.arange(N // 2 + 1)
    b = np.where(k == 0, 1.0, 2.0 / (1.0 - 4.0 * k * k))
    if N % 2 == 0:
        b[-1] *= 0.5
    j = np.arange(M)
    w = 2.0 * (np.cos(2.0 * np.pi * np.outer(j, k) / N) @ b) / N
    w[0] *= 0.5
    w[-1] *= 0.5
    return w


def interp_inner_to_full(x_inner: np.ndarray, x_full: np.ndarray) -> np.ndarray:
    """Degree-(Mi-1) polynomial through the inner nodes, evaluated on all nodes (sg.py:212-248)."""
    a, b = x_full[0], x_full[-1]
    xi_in = 2 * (x_inner - a) / (b - a) - 1
    xi_fu = 2 * (x_full - a) / (b - a) - 1
    n = x_inner.size
    V_in = chebvander(xi_in, n - 1)
    V_fu = chebvander(xi_fu, n - 1)
    return V_fu @ np.linalg.solve(V_in, np.eye(n))


# --------------------------------------------------------------------------- Legendre (basis_type="legendre")
def lgl_nodes(M: int) -> np.ndarray:
    """Legendre-Gauss-Lobatto nodes: +-1 and the roots of P_N' (basis/polynomial.py:164-195)."""
    from numpy.polynomial.legendre import Legendre
    roots = Legendre.basis(M - 1).deriv().roots()
    return np.sort(np.concatenate(([-1.0], roots, [1.0])))


def lgl_weights(M: int) -> np.ndarray:
    """w_j = 2 / (N (N+1) P_N(x_j)^2) (basis/polynomial.py:198-243)."""
    N = M - 1
    if N == 0:
        return np.array([2.0])
    PN = jacobi_p(lgl_nodes(M), 0.0, 0.0, N)
    return 2.0 / (N * (N + 1) * PN**2)


def legendre_diff_reference_interval(xi: np.ndarray) -> np.ndarray:
    """D = Vx V^-1 with V_jn = P_n(x_j), Vx_jn = P_n'(x_j) = (n+1)/2 P^{(1,1)}_{n-1}(x_j)
    (basis/spectral.py:93-130, basis/polynomial.py:132-157, 250-345)."""
    n = xi.size
    V = np.zeros((n, n))
    Vx = np.zeros((n, n))
    for k in range(n):
        V[:, k] = jacobi_p(xi, 0.0, 0.0, k)
        Vx[:, k] = 0.0 if k == 0 else 0.5 * (k + 1) * jacobi_p(xi, 1.0, 1.0, k - 1)
    return Vx @ np.linalg.solve(V, np.eye(n))


@dataclass
class Axis:
    """Everything one coordinate direction needs (sg.py:103-119, 181-210, 479-493)."""
    N: int
    L: float
    kind: str = "chebyshev"
    x: np.ndarray = field(init=False)
    D: np.ndarray = field(init=False)
    D2: np.ndarray = field(init=False)
    I: np.ndarray = field(init=False)   # (M, Mi) inner -> full interpolation
    w: np.ndarray = field(init=False)
    hmin: float = field(init=False)

    def __post_init__(self):
        M = self.N + 1
        if self.kind == "legendre":                            # basis/spectral.py:326-407
            xi = lgl_nodes(M)
            Dxi, wxi = legendre_diff_reference_interval(xi), lgl_weights(M)
        else:
            xi = cgl_nodes(M)
            Dxi, wxi = cheb_diff_reference_interval(xi), clenshaw_curtis(M)
        self.x = 0.5 * (self.L - 0.0) * (xi + 1.0) + 0.0      # basis/spectral.py:498-502
        self.D = (2.0 / (self.L - 0.0)) * Dxi                  # :518-522
        self.D2 = self.D @ self.D                              # sg.py:192-193
        self.I = interp_inner_to_full(self.x[1:-1], self.x)    # sg.py:209-210
        self.w = wxi * (self.L - 0.0) / 2                      # basis/spectral.py:538-541
        self.hmin = float(np.min(np.diff(self.x)))             # sg.py:118-119


# --------------------------------------------------------------------------- lid profile
def lid_profile(x: np.ndarray, method: str, width: float, U: float, Lx: float) -> np.ndarray:
    """u on the lid (operators/corner.py:80-112 smoothing, :148-169 Saad, :192-223 factory)."""
    m = method.lower()
    x = np.asarray(x, dtype=float)
    if m == "smoothing":
        u = np.full_like(x, U)
        if width > 0:
            d = width * Lx
            left = x < d
            u[left] = 0.5 * (1 - np.cos(np.pi * x[left] / d)) * U
            right = x > (Lx - d)
            u[right] = 0.5 * (1 - np.cos(np.pi * (Lx - x[right]) / d)) * U
        return u
    if m in ("saad", "polynomial"):
        s = x / Lx
        return 16.0 * s**2 * (1.0 - s) ** 2 * U
    raise ValueError(
        f"Unknown corner treatment method: {method}. Use 'smoothing', 'polynomial', or 'saad'."
    )


# --------------------------------------------------------------------------- the solver
class OracleSG:
    """NumPy restatement of ``SGSolver`` + the ``LidDrivenCavitySolver.solve`` loop.

    ``stage_pressure=False`` reproduces quirk Q1 (SG always differentiates p^n,
    sg.py:270/329); ``True`` gives the FSG-smoother behaviour (multigrid/fsg.py:880).
    """

    def __init__(self, N, Re, *, lid_velocity=1.0, Lx=1.0, Ly=1.0, CFL=1.5, beta_squared=5.0,
                 corner_treatment="smoothing", corner_smoothing=0.15, stage_pressure=False,
                 basis_type="chebyshev", ny=None):
        # (ny: polynomial order of the y grid when it differs from N = nx -- reference sg.py:103-119 builds the two
        #  grids independently)
        self.N, self.Re = int(N), float(Re)
        self.Ny = int(N if ny is None else ny)
        self.U, self.Lx, self.Ly = float(lid_velocity), float(Lx), float(Ly)
        self.CFL, self.beta2 = float(CFL), float(beta_squared)
        self.stage_pressure = bool(stage_pressure)
        kind = basis_type.lower()
        if kind not in ("chebyshev", "legendre"):                # sg.py:52-63
            raise ValueError(f"Unknown basis_type: {basis_type}. Use 'legendre' or 'chebyshev'")
        self.ax = Axis(self.N, self.Lx, kind)
        self.ay = Axis(self.Ny, self.Ly, kind)
        M, My = self.N + 1, self.Ny + 1
        self.M, self.Mi = M, M - 2
        self.My, self.Myi = My, My - 2
        self.u_lid = lid_profile(self.ax.x, corner_treatment, corner_smoothing, self.U, self.Lx)
        self.W = np.outer(self.ax.w, self.ay.w)                # sg.py:493
        self.u = np.zeros((M, My))
        self.v = np.zeros((M, My))
        self.p = np.zeros((self.Mi, self.Myi))
        self.u[:, -1] = self.u_lid                             # sg.py:98, 250-253
        self.Ru = np.zeros((M, My))
        self.Rv = np.zeros((M, My))
        self.Rp = np.zeros((self.Mi, self.Myi))

    # -- a8: sg.py:387-408
    def timestep(self) -> float:
        umax = max(np.max(np.abs(self.u)), self.U)
        vmax = max(np.max(np.abs(self.v)), 1e-10)
        nu = 1.0 / self.Re
        hx, hy = self.ax.hmin, self.ay.hmin
        lam_x = (umax + np.sqrt(umax**2 + self.beta2)) / hx + nu / hx**2
        lam_y = (vmax + np.sqrt(vmax**2 + self.beta2)) / hy + nu / hy**2
        return self.CFL / (lam_x + lam_y)

    # -- a10: sg.py:255-276
    def pressure_gradient(self, p):
        pf = self.ax.I @ p @ self.ay.I.T
        return self.ax.D @ pf, pf @ self.ay.D.T

    # -- a9: sg.py:278-346
    def residual(self, u, v, p, want_parts=False):
        Dx, Dy, D2x, D2y = self.ax.D, self.ay.D, self.ax.D2, self.ay.D2
        ux, uy = Dx @ u, u @ Dy.T
        vx, vy = Dx @ v, v @ Dy.T
        lap_u = D2x @ u + u @ D2y.T
        lap_v = D2x @ v + v @ D2y.T
        px, py = self.pressure_gradient(p if self.stage_pressure else self.p)
        nu = 1.0 / self.Re
        Ru = -(u * ux + v * uy) - px + nu * lap_u
        Rv = -(u * vx + v * vy) - py + nu * lap_v
        Rp = -self.beta2 * (ux + vy)[1:-1, 1:-1]
        if want_parts:
            return Ru, Rv, Rp, dict(du_dx=ux, du_dy=uy, dv_dx=vx, dv_dy=vy, lap_u=lap_u,
                                    lap_v=lap_v, dp_dx=px, dp_dy=py)
        return Ru, Rv, Rp

    # -- a12: sg.py:348-385 (walls first, lid last so the top corners take the lid value)
    def apply_bc(self, u, v):
        u[0, :] = 0.0
        v[0, :] = 0.0
        u[-1, :] = 0.0
        v[-1, :] = 0.0
        u[:, 0] = 0.0
        v[:, 0] = 0.0
        u[:, -1] = self.u_lid
        v[:, -1] = 0.0

    # -- a11: sg.py:410-449
    def step(self) -> float:
        dt = self.timestep()
        u0, v0, p0 = self.u, self.v, self.p
        ui, vi, pi = u0, v0, p0
        for a in RK_ALPHAS:
            Ru, Rv, Rp = self.residual(ui, vi, pi)
            us = u0 + a * dt * Ru
            vs = v0 + a * dt * Rv
            ps = p0 + a * dt * Rp
            self.apply_bc(us, vs)
            ui, vi, pi = us, vs, ps
        self.u, self.v, self.p = ui, vi, pi
        self.Ru, self.Rv, self.Rp = Ru, Rv, Rp     # last-stage residual (quirk Q4)
        return dt

    # -- a14: sg.py:463-473
    def residual_norms(self):
        return (float(np.linalg.norm(self.Ru)), float(np.linalg.norm(self.Rv)),
                float(np.linalg.norm(self.Rp)))

    # -- a15: sg.py:495-550
    def vorticity(self):
        return self.ax.D @ self.v - self.u @ self.ay.D.T

    def energy(self):
        return 0.5 * float(np.sum(self.W * (self.u * self.u + self.v * self.v)))

    def enstrophy(self):
        w = self.vorticity()
        return 0.5 * float(np.sum(self.W * w * w))

    def palinstrophy(self):
        w = self.vorticity()
        wx, wy = self.ax.D @ w, w @ self.ay.D.T
        return 0.5 * float(np.sum(self.W * (wx**2 + wy**2)))

    # -- a13: base.py:202-330
    def solve(self, tolerance=1e-6, max_iter=10_000_000, diagnostics=True, nan_guard=False):
        hist = dict(rel=[], ru=[], rv=[], rp=[], E=[], Z=[], P=[], dt=[])
        up, vp = self.u.copy(), self.v.copy()
        its, conv = 0, False
        for i in range(max_iter):
            its = i + 1
            dt = self.step()
            du = np.linalg.norm(self.u - up) / (np.linalg.norm(up) + 1e-12)
            dv = np.linalg.norm(self.v - vp) / (np.linalg.norm(vp) + 1e-12)
            rel = max(du, dv)
            if i >= 10:
                ru, rv, rp = self.residual_norms()
                hist["rel"].append(rel); hist["ru"].append(ru); hist["rv"].append(rv)
                hist["rp"].append(rp); hist["dt"].append(dt)
                if diagnostics:
                    hist["E"].append(self.energy()); hist["Z"].append(self.enstrophy())
                    hist["P"].append(self.palinstrophy())
            up, vp = self.u.copy(), self.v.copy()
            conv = (i >= 10) and (rel < tolerance)
            if conv:
                break
            if nan_guard and not math.isfinite(rel):
                break
        return its, conv, hist

    # -- a16: sg.py:144-179, 451-461 (quirk Q5: linear extrapolation, not the interpolant)
    def pressure_on_full_grid(self):
        f = np.zeros((self.M, self.My))
        f[1:-1, 1:-1] = self.p
        f[0, 1:-1] = 2 * f[1, 1:-1] - f[2, 1:-1]
        f[-1, 1:-1] = 2 * f[-2, 1:-1] - f[-3, 1:-1]
        f[1:-1, 0] = 2 * f[1:-1, 1] - f[1:-1, 2]
        f[1:-1, -1] = 2 * f[1:-1, -2] - f[1:-1, -3]
        f[0, 0] = 0.5 * (f[0, 1] + f[1, 0])
        f[0, -1] = 0.5 * (f[0, -2] + f[1, -1])
        f[-1, 0] = 0.5 * (f[-1, 1] + f[-2, 0])
        f[-1, -1] = 0.5 * (f[-1, -2] + f[-2, -1])
        return f

    # -- a17: sg.py:556-619.  The reference assembles the Kronecker system with identity
    #    rows on the boundary and calls SuperLU; with psi = 0 on the walls that is exactly
    #    the interior Sylvester problem A Psi + Psi B^T = -Omega_int, A = D2x[1:-1,1:-1].
    def streamfunction(self):
        from scipy.linalg import solve_sylvester
        w = self.vorticity()
        A = self.ax.D2[1:-1, 1:-1]
        B = self.ay.D2[1:-1, 1:-1]
        psi = np.zeros((self.M, self.My))
        psi[1:-1, 1:-1] = solve_sylvester(A, B.T, -w[1:-1, 1:-1])
        return psi

    # -- a18: sg.py:621-743
    def vortex_metrics(self, psi=None):
        psi = self.streamfunction() if psi is None else psi
        w = self.vorticity()
        X, Y = np.meshgrid(self.ax.x, self.ay.x, indexing="ij")
        out = {}
        k = np.unravel_index(np.argmin(psi), psi.shape)
        out.update(psi_min=float(psi[k]), psi_min_x=float(X[k]), psi_min_y=float(Y[k]),
                   omega_center=float(w[k]))
        k = np.unravel_index(np.argmax(np.abs(w)), w.shape)
        out.update(omega_max=float(w[k]), omega_max_x=float(X[k]), omega_max_y=float(Y[k]))
        regions = {"BR": (X > 0.5) & (Y < 0.5), "BL": (X < 0.5) & (Y < 0.5),
                   "TL": (X < 0.5) & (Y > 0.5)}
        for name, mask in regions.items():
            k = np.unravel_index(np.argmax(np.where(mask, psi, -np.inf)), psi.shape)
            if psi[k] > 0:
                vals = (float(psi[k]), float(w[k]), float(X[k]), float(Y[k]))
            else:
                vals = (0.0, 0.0, 0.0, 0.0)
            out[f"psi_{name}"], out[f"omega_{name}"], out[f"psi_{name}_x"], out[f"psi_{name}_y"] = vals
        return out


# --------------------------------------------------------------------------- Ghia metric
def jacobi_p(x, alpha, beta, n):
    """P_n^{(alpha,beta)}(x) by the three-term recurrence (basis/polynomial.py:15-73)."""
    x = np.asarray(x, dtype=float)
    p_prev = np.ones_like(x)
    if n == 0:
        return p_prev
    p_cur = 0.5 * (alpha - beta + (alpha + beta + 2) * x)
    for m in range(1, n):
        s = 2 * m + alpha + beta
        a_lo = 2 * (m + alpha) * (m + beta) / ((s + 1) * s)
        a_mid = (alpha**2 - beta**2) / ((s + 2) * s) if (alpha != beta) else 0.0
        a_hi = 2 * (m + 1) * (m + alpha + beta + 1) / ((s + 2) * (s + 1))
        p_prev, p_cur = p_cur, ((a_mid + x) * p_cur - a_lo * p_prev) / a_hi
    return p_cur


def spectral_interpolate(x_nodes, f, x_eval, basis="legendre"):
    """Modal (Vandermonde) interpolation used for centreline extraction (polynomial.py:398-477)."""
    ab = {"legendre": (0.0, 0.0), "chebyshev": (-0.5, -0.5)}
    if basis.lower() not in ab:
        raise ValueError(f"Unknown basis: {basis}. Use 'legendre' or 'chebyshev'.")
    al, be = ab[basis.lower()]
    lo, hi = x_nodes.min(), x_nodes.max()
    if not (np.isclose(lo, -1.0) and np.isclose(hi, 1.0)):
        xn = 2.0 * (x_nodes - lo) / (hi - lo) - 1.0
        xe = 2.0 * (np.asarray(x_eval) - lo) / (hi - lo) - 1.0
    else:
        xn, xe = x_nodes, np.asarray(x_eval)
    n = len(x_nodes)
    V = np.stack([jacobi_p(xn, al, be, k) for k in range(n)], axis=1)
    Ve = np.stack([jacobi_p(xe, al, be, k) for k in range(n)], axis=1)
    return Ve @ np.linalg.solve(V, f)


def ghia_centerline_error(x, y, U, V, ghia_u_xy, ghia_v_xy):
    """RMS / relative L2 centreline error at the Ghia points (recipe of
    src/shared/plotting/ldc/validation.py:297-322: column nearest x = centre of U along y,
    row nearest y = centre of V along x, Legendre-modal interpolation)."""
    ic = int(np.argmin(np.abs(x - 0.5 * (x.min() + x.max()))))
    jc = int(np.argmin(np.abs(y - 0.5 * (y.min() + y.max()))))
    yu, ug = np.asarray(ghia_u_xy[0], float), np.asarray(ghia_u_xy[1], float)
    xv, vg = np.asarray(ghia_v_xy[0], float), np.asarray(ghia_v_xy[1], float)
    ui = spectral_interpolate(y, U[ic, :], yu)
    vi = spectral_interpolate(x, V[:, jc], xv)
    eu, ev = ui - ug, vi - vg
    return dict(u_rms=float(np.sqrt(np.mean(eu**2))), v_rms=float(np.sqrt(np.mean(ev**2))),
                u_rel=float(np.linalg.norm(eu) / np.linalg.norm(ug)),
                v_rel=float(np.linalg.norm(ev) / np.linalg.norm(vg)))


# --------------------------------------------------------------------------- FSG (a20)
def fft_prolongation_matrix(nc: int, nf: int) -> np.ndarray:
    """(nf, nc) matrix of the reference's ``FFTProlongation.prolongate_1d``
    (operators/transfer_operators.py:209-255), restated in closed form.

    The reference halves the end samples, applies SciPy's un-normalised DCT-I (which already
    weights the ends by 1/2 relative to the interior: y_k = x_0 + (-1)^k x_N + 2 sum x_n cos),
    divides by N_c, halves the end coefficients and sums c_k cos(k pi i / N_f).  The double end
    weighting makes it NOT an interpolation (quirk Q10); it is reproduced as is.  It is linear,
    so the matrix is the operator applied to unit vectors."""
    if nc == nf:
        return np.eye(nc)
    if nc > nf:
        raise ValueError(f"Prolongation requires n_coarse ({nc}) <= n_fine ({nf})")
    Nc, Nf = nc - 1, nf - 1
    k = np.arange(nc)
    j = np.arange(nc)
    dct1 = 2.0 * np.cos(np.pi * np.outer(k, j) / Nc)      # interior columns
    dct1[:, 0] = 1.0
    dct1[:, -1] = (-1.0) ** k
    w_in = np.ones(nc); w_in[[0, -1]] = 0.5                # "u_weighted[0] /= 2 ..."
    w_c = np.ones(nc); w_c[[0, -1]] = 0.5                  # "coeffs[0] /= 2 ..."
    coeff = (w_c[:, None] * dct1 * w_in[None, :]) / Nc     # coefficients of each unit vector
    ev = np.cos(np.pi * np.outer(np.arange(nf), k) / Nf)   # T_k at the fine angles
    return ev @ coeff


def polynomial_prolongation_matrix(nc: int, nf: int) -> np.ndarray:
    """(nf, nc) matrix of the reference's ``PolynomialProlongation.prolongate_1d``
    (operators/transfer_operators.py:333-376): the degree-(nc-1) Chebyshev interpolant of the coarse CGL data
    evaluated on the fine CGL nodes.  The reference fits with ``chebfit`` on cos(pi k / N) (descending); index k
    of the coarse grid maps to index i of the fine grid the same way on the ascending nodes used here (mirror
    symmetry), so the matrix is the same."""
    if nc == nf:
        return np.eye(nc)
    xc = np.cos(np.pi * np.arange(nc) / (nc - 1))
    xf = np.cos(np.pi * np.arange(nf) / (nf - 1))
    return chebvander(xf, nc - 1) @ np.linalg.solve(chebvander(xc, nc - 1), np.eye(nc))


def prolongation_matrix(method: str, nc: int, nf: int) -> np.ndarray:
    """create_transfer_operators (operators/transfer_operators.py:503-529)."""
    if method == "fft":
        return fft_prolongation_matrix(nc, nf)
    if method == "polynomial":
        return polynomial_prolongation_matrix(nc, nf)
    raise ValueError(f"Unknown prolongation method: {method}")


def fsg_orders(n_fine: int, n_levels: int, coarsest_n: int = 12) -> list:
    """Polynomial orders coarse -> fine (multigrid/fsg.py:517-531)."""
    orders, n = [], n_fine
    for _ in range(n_levels):
        orders.append(n)
        if n // 2 < coarsest_n:
            break
        n //= 2
    return orders[::-1]


def fsg_prolongate(coarse: "OracleSG", fine: "OracleSG", lid_velocity: float, method: str = "fft"):
    """multigrid/fsg.py:551-614 including quirk Q2: the boundary re-imposition uses [ix, iy]
    arrays as if they were [iy, ix], so the EAST wall gets the (scalar) lid speed and the lid
    row is zeroed; the caller's initialize_lid then restores the lid column."""
    Pf = prolongation_matrix(method, coarse.M, fine.M)
    Pi = prolongation_matrix(method, coarse.Mi, fine.Mi)
    u = Pf @ coarse.u @ Pf.T
    v = Pf @ coarse.v @ Pf.T
    u[0, :] = 0.0; v[0, :] = 0.0
    u[-1, :] = lid_velocity; v[-1, :] = 0.0
    u[:, 0] = 0.0; v[:, 0] = 0.0
    u[:, -1] = 0.0; v[:, -1] = 0.0
    fine.u, fine.v = u, v
    fine.p = Pi @ coarse.p @ Pi.T


def oracle_fsg(N, Re, *, tolerance=1e-6, max_iterations=500000, n_levels=2, coarse_tolerance_factor=1.0,
               prolongation_method="fft", **kw):
    """``solve_fsg`` (multigrid/fsg.py:1053-1221): coarse -> fine, the smoother differentiates the
    STAGE pressure, no warm-up, NaN/Inf exit.  Returns (finest level, total iterations, converged)."""
    orders = fsg_orders(N, n_levels)
    U = kw.get("lid_velocity", 1.0)
    levels = [OracleSG(n, Re, stage_pressure=True, **kw) for n in orders]
    total, converged, diverged = 0, False, False
    for idx, lvl in enumerate(levels):
        tol = tolerance * coarse_tolerance_factor ** (len(levels) - 1 - idx)
        if idx == 0:
            lvl.u[:] = 0.0; lvl.v[:] = 0.0; lvl.p[:] = 0.0
        else:
            fsg_prolongate(levels[idx - 1], lvl, U, prolongation_method)
        lvl.u[:, -1] = lvl.u_lid                       # initialize_lid (:950-954)
        lvl.v[:, -1] = 0.0
        converged = False
        for _ in range(max_iterations):
            up, vp = lvl.u.copy(), lvl.v.copy()
            lvl.step()
            ur = np.linalg.norm(lvl.u - up) / (np.linalg.norm(up) + 1e-12)
            vr = np.linalg.norm(lvl.v - vp) / (np.linalg.norm(vp) + 1e-12)
            total += 1
            mr = max(ur, vr)
            if mr < tol:
                converged = True
                break
            if not np.isfinite(mr):
                diverged = True
                break
        if diverged:
            break
    return levels[-1], total, bool(converged and not diverged)
