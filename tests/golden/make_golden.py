#!/usr/bin/env python3
"""Generate golden vectors by running the *reference* spectral solver in this container.

TEST INFRASTRUCTURE.  This script is the only place that touches ``/root/reference``.
It imports ``solvers.spectral.sg`` from there (read-only, by path), runs it on small
deterministic cases and writes ``.npz`` / ``.json`` fixtures next to itself.  Only the
numbers it writes are committed; no reference source is copied.

The reference cannot be imported as-is here (hydra/mlflow/pyvista/numba are not
installed), so three *harness-side* stand-ins are registered before the import
(recipe of SURVEY.md section 8c): a ``numba`` whose ``njit`` is the identity
decorator, an ``mlflow`` with ``active_run() -> None`` and a ``pyvista`` placeholder;
and an empty ``solvers`` package whose ``__path__`` points into the reference so that
``solvers/__init__.py`` (which pulls in the FV solver) is skipped.  ``numpy.kron`` is
guarded while constructing solvers, because ``sg.py:197-205`` builds five dense
(N+1)^2 x (N+1)^2 Kronecker matrices that are never read again.

Usage:  python tests/golden/make_golden.py [--only G1,G4] [--full]
"""
from __future__ import annotations

import argparse
import contextlib
import importlib
import json
import sys
import time
import types
from pathlib import Path

import numpy as np

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent


# --------------------------------------------------------------------------- shims
def _install_shims():
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")

        def njit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda f: f

        nb.njit = njit
        nb.jit = njit
        nb.prange = range
        sys.modules["numba"] = nb
    if "mlflow" not in sys.modules:
        ml = types.ModuleType("mlflow")
        ml.active_run = lambda: None
        ml.log_metrics = lambda *a, **k: None
        sys.modules["mlflow"] = ml
    if "pyvista" not in sys.modules:
        pv = types.ModuleType("pyvista")

        class StructuredGrid:  # placeholder: only the annotation in base.py needs it
            pass

        pv.StructuredGrid = StructuredGrid

        def _read(path):
            """Harness-side stand-in for pyvista.read: decode the .vts with the stdlib reader of
            this repository and expose the two attributes base.py:1008-1015 uses."""
            import importlib.util
            spec = importlib.util.spec_from_file_location(
                "ldc_vtkio", OUT.parent.parent / "02689-advancednumericalalgorithmp3_amd" / "src" / "solvers" / "vtkio.py")
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            g = mod.read_vts(path)
            return types.SimpleNamespace(point_data=g["point_data"], points=g["points"])

        pv.read = _read
        sys.modules["pyvista"] = pv
    src = str(REF / "src")
    if src not in sys.path:
        sys.path.insert(0, src)
    if "solvers" not in sys.modules:
        pkg = types.ModuleType("solvers")
        pkg.__path__ = [str(REF / "src" / "solvers")]
        sys.modules["solvers"] = pkg


@contextlib.contextmanager
def kron_guard(max_rows: int = 4096):
    """Replace np.kron by a stub for huge (dead-store) products during construction."""
    real = np.kron

    def guarded(a, b):
        a = np.asarray(a)
        b = np.asarray(b)
        if a.shape[0] * b.shape[0] > max_rows:
            return np.zeros((1, 1))
        return real(a, b)

    np.kron = guarded
    try:
        yield
    finally:
        np.kron = real


def ref_modules():
    _install_shims()
    sg = importlib.import_module("solvers.spectral.sg")
    return sg


def make_sg(N, Re, **kw):
    sg = ref_modules()
    args = dict(
        name="spectral", Re=Re, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N,
        tolerance=1e-6, max_iterations=10_000_000, basis_type="chebyshev", CFL=1.5,
        beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
        multigrid="none",
    )
    args.update(kw)
    with kron_guard():
        return sg.SGSolver(**args)


# --------------------------------------------------------------------------- groups
def g1_operators():
    """Nodes, D, D2, pressure interpolation, quadrature weights (a1-a5)."""
    out = {}
    for N in (8, 16, 33, 64):
        s = make_sg(N, 100.0)
        out[f"N{N}_x"] = s.basis_x.nodes(N + 1)
        out[f"N{N}_Dx"] = s.Dx_1d
        out[f"N{N}_Dxx"] = s.Dxx_1d
        out[f"N{N}_Interp_x"] = s.Interp_x
        out[f"N{N}_w_x"] = s.w_x
        out[f"N{N}_dx_min"] = np.array(s.dx_min)
    # non-unit domain: checks the 2/L scaling
    s = make_sg(12, 100.0, Lx=2.0, Ly=0.5)
    out["L_x"] = s.basis_x.nodes(13)
    out["L_y"] = s.basis_y.nodes(13)
    out["L_Dx"] = s.Dx_1d
    out["L_Dy"] = s.Dy_1d
    out["L_w_x"] = s.w_x
    out["L_w_y"] = s.w_y
    out["L_Interp_y"] = s.Interp_y
    np.savez_compressed(OUT / "g1_operators.npz", **out)


def g2_lid():
    """Lid profiles for several corner settings (a6)."""
    out = {}
    for cs in (0.0, 0.01, 0.15, 0.35, 0.5):
        s = make_sg(32, 100.0, corner_smoothing=cs)
        out[f"smooth_{cs}"] = s.u_2d[:, -1].copy()
    s = make_sg(32, 100.0, corner_treatment="saad")
    out["saad"] = s.u_2d[:, -1].copy()
    s = make_sg(20, 100.0, corner_smoothing=0.2, lid_velocity=2.5, Lx=2.0)
    out["smooth_L2_U2.5"] = s.u_2d[:, -1].copy()
    np.savez_compressed(OUT / "g2_lid.npz", **out)


def _smooth_state(N, seed=0):
    """Deterministic smooth (low-degree polynomial) fields on the N+1 grid."""
    rng = np.random.default_rng(seed)
    M = N + 1
    x = 0.5 * (1.0 - np.cos(np.pi * np.arange(M) / N))
    X, Y = np.meshgrid(x, x, indexing="ij")

    def poly():
        c = rng.standard_normal((5, 5))
        return sum(c[a, b] * X**a * Y**b for a in range(5) for b in range(5))

    return poly(), poly(), poly()[1:-1, 1:-1].copy()


def g3_single_stage():
    """One residual evaluation from a fixed smooth state (a9-a10)."""
    out = {}
    for N in (16, 24):
        s = make_sg(N, 400.0)
        u, v, p = _smooth_state(N, seed=N)
        s.arrays.u[:] = u.ravel()
        s.arrays.v[:] = v.ravel()
        s.arrays.p[:] = p.ravel()
        s._compute_residuals(s.arrays.u, s.arrays.v, s.arrays.p)
        a = s.arrays
        for k in ("u", "v", "p", "du_dx", "du_dy", "dv_dx", "dv_dy", "lap_u", "lap_v",
                  "dp_dx", "dp_dy", "R_u", "R_v", "R_p"):
            out[f"N{N}_{k}"] = getattr(a, k).copy()
    np.savez_compressed(OUT / "g3_single_stage.npz", **out)


def _run_steps(s, K, record_every=None):
    """Drive the reference exactly as base.solve() does, for K iterations."""
    dts, rel, res = [], [], []
    E, Z, P = [], [], []
    u_prev = s.arrays.u.copy()
    v_prev = s.arrays.v.copy()
    for i in range(K):
        dts.append(s._compute_adaptive_timestep())
        s.arrays.u, s.arrays.v, s.arrays.p = s.step()
        du = np.linalg.norm(s.arrays.u - u_prev) / (np.linalg.norm(u_prev) + 1e-12)
        dv = np.linalg.norm(s.arrays.v - v_prev) / (np.linalg.norm(v_prev) + 1e-12)
        rel.append(max(du, dv))
        r = s._compute_algebraic_residuals()
        res.append([r["u_residual"], r["v_residual"], r["continuity_residual"]])
        E.append(s._compute_energy())
        Z.append(s._compute_enstrophy())
        P.append(s._compute_palinstrophy())
        u_prev = s.arrays.u.copy()
        v_prev = s.arrays.v.copy()
    return dict(dt=np.array(dts), rel=np.array(rel), res=np.array(res),
                E=np.array(E), Z=np.array(Z), P=np.array(P))


def g4_trajectories(full=False):
    """K-step trajectories from rest (a8, a11-a15; exposes quirk Q1)."""
    cases = [(16, 100.0, 50), (32, 100.0, 500), (64, 400.0, 1000)]
    if full:
        cases.append((64, 1000.0, 3000))
    for N, Re, K in cases:
        t0 = time.time()
        s = make_sg(N, Re)
        h = _run_steps(s, K)
        out = dict(u=s.arrays.u.copy(), v=s.arrays.v.copy(), p=s.arrays.p.copy(),
                   R_u=s.arrays.R_u.copy(), R_v=s.arrays.R_v.copy(), R_p=s.arrays.R_p.copy(),
                   omega=s._compute_vorticity(), **h)
        # psi + vortex metrics on the end state through the reference's spsolve (a17-a18)
        if N <= 64:
            psi, _, _ = s._compute_streamfunction()
            out["psi"] = psi
            vm = s.compute_vortex_metrics()
            out["vortex_keys"] = np.array(sorted(vm))
            out["vortex_vals"] = np.array([vm[k] for k in sorted(vm)])
        np.savez_compressed(OUT / f"g4_traj_N{N}_Re{int(Re)}_K{K}.npz", **out)
        print(f"  traj N={N} Re={Re} K={K}: {time.time() - t0:.1f}s")


def _compact_state(a, name, out):
    """A 2-D array in compact form: six rows and six columns (first two, quarter, middle, last two), the lattice of
    every 8th node, the 2-norm and the maximum -- a few KB that pin every node class of a large grid."""
    n0, n1 = a.shape
    rows = sorted({0, 1, n0 // 4, n0 // 2, n0 - 2, n0 - 1})
    cols = sorted({0, 1, n1 // 4, n1 // 2, n1 - 2, n1 - 1})
    out[f"{name}_rows_idx"] = np.array(rows)
    out[f"{name}_cols_idx"] = np.array(cols)
    out[f"{name}_rows"] = a[rows, :].copy()
    out[f"{name}_cols"] = a[:, cols].copy()
    out[f"{name}_lattice"] = a[::8, ::8].copy()
    out[f"{name}_norm2"] = np.array(np.linalg.norm(a))
    out[f"{name}_max"] = np.array(np.max(np.abs(a)))


def g4c_headline_sizes():
    """The reference itself at the two large BASELINE geometries (configs 3-5): K steps from rest at (N=128, Re=1000, K=40)
    and (N=256, Re=1000, K=25), and ten / eight times as long (K=400, K=200) -- every record in full, the end state in compact form (_compact_state)."""
    for N, Re, K in ((128, 1000.0, 40), (256, 1000.0, 25), (128, 1000.0, 400), (256, 1000.0, 200)):
        t0 = time.time()
        s = make_sg(N, Re)
        h = _run_steps(s, K)
        M = N + 1
        out = dict(**h)
        _compact_state(s.arrays.u.reshape(M, M), "u", out)
        _compact_state(s.arrays.v.reshape(M, M), "v", out)
        _compact_state(s.arrays.p.reshape(M - 2, M - 2), "p", out)
        _compact_state(np.asarray(s._compute_vorticity()).reshape(M, M), "omega", out)
        np.savez_compressed(OUT / f"g4c_traj_N{N}_Re{int(Re)}_K{K}.npz", **out)
        print(f"  traj N={N} Re={Re} K={K}: {time.time() - t0:.1f}s")


def g4b_variants():
    """Short trajectories with non-default parameters (Saad lid, other CFL/beta, Lx!=Ly)."""
    cases = {
        "saad": dict(N=24, Re=100.0, K=60, kw=dict(corner_treatment="saad")),
        "cfl": dict(N=20, Re=50.0, K=60, kw=dict(CFL=0.8, beta_squared=2.0, corner_smoothing=0.05)),
        "rect": dict(N=18, Re=100.0, K=40, kw=dict(Lx=2.0, Ly=1.0, lid_velocity=1.5)),
        "odd": dict(N=15, Re=100.0, K=80, kw=dict()),
    }
    out = {}
    for name, c in cases.items():
        s = make_sg(c["N"], c["Re"], **c["kw"])
        h = _run_steps(s, c["K"])
        out[f"{name}_u"] = s.arrays.u.copy()
        out[f"{name}_v"] = s.arrays.v.copy()
        out[f"{name}_p"] = s.arrays.p.copy()
        for k, val in h.items():
            out[f"{name}_{k}"] = val
    np.savez_compressed(OUT / "g4b_variants.npz", **out)
    meta = {k: dict(N=c["N"], Re=c["Re"], K=c["K"], kw=c["kw"]) for k, c in cases.items()}
    (OUT / "g4b_variants.json").write_text(json.dumps(meta, indent=1))


def g13_unequal_grids():
    """nx != ny (reference sg.py:103-119 builds the two grids independently; no shipped config uses it): short
    trajectories from rest with the full set of records, omega, psi and the vortex metrics of the end state."""
    cases = {
        "nx24_ny40": dict(nx=24, ny=40, Re=100.0, K=60, kw=dict()),
        "nx40_ny20": dict(nx=40, ny=20, Re=400.0, K=80, kw=dict(Lx=2.0, Ly=1.0)),
        "nx17_ny32": dict(nx=17, ny=32, Re=100.0, K=60, kw=dict(corner_treatment="saad")),
    }
    out, meta = {}, {}
    for name, c in cases.items():
        s = make_sg(c["nx"], c["Re"], ny=c["ny"], **c["kw"])
        h = _run_steps(s, c["K"])
        out[f"{name}_u"] = s.arrays.u.copy()
        out[f"{name}_v"] = s.arrays.v.copy()
        out[f"{name}_p"] = s.arrays.p.copy()
        out[f"{name}_omega"] = s._compute_vorticity()
        psi, _, _ = s._compute_streamfunction()
        out[f"{name}_psi"] = psi
        vm = s.compute_vortex_metrics()
        meta[name] = dict(nx=c["nx"], ny=c["ny"], Re=c["Re"], K=c["K"], kw=c["kw"],
                          vortex={k: float(v) for k, v in vm.items()})
        for k, val in h.items():
            out[f"{name}_{k}"] = val
        print(f"  {name}: u {s.arrays.u.shape}, p {s.arrays.p.shape}, psi {np.shape(psi)}")
    np.savez_compressed(OUT / "g13_unequal_grids.npz", **out)
    (OUT / "g13_unequal_grids.json").write_text(json.dumps(meta, indent=1))


def g7_converged(N=32, Re=100.0):
    """Full solve() at (N, Re), tol 1e-6 through the reference's own loop (a13, a16)."""
    t0 = time.time()
    tag = f"g7_converged_N{N}_Re{int(Re)}"
    s = make_sg(N, Re)
    s.solve()
    m = {k: (v.item() if isinstance(v, np.generic) else v)
         for k, v in s.metrics.__dict__.items()}
    m["wall_time_seconds"] = float(m["wall_time_seconds"])
    ts = {k: list(map(float, v)) if v else [] for k, v in s.time_series.__dict__.items()}
    # FV comparison through the reference's own compute_validation_errors (CWD = reference root,
    # because base.py:995-1001 uses relative data/validation paths)
    import os
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        m["validation_errors"] = s.compute_validation_errors(save_plots=False)
    finally:
        os.chdir(cwd)
    (OUT / f"{tag}.json").write_text(
        json.dumps(dict(metrics=m, time_series_len={k: len(v) for k, v in ts.items()},
                        time_series_head={k: v[:5] for k, v in ts.items()},
                        time_series_tail={k: v[-5:] for k, v in ts.items()}), indent=1))
    np.savez_compressed(OUT / f"{tag}.npz",
                        u=s.fields.u, v=s.fields.v, p=s.fields.p, x=s.fields.x, y=s.fields.y,
                        p_inner=s.arrays.p,
                        **{f"ts_{k}": np.array(v) for k, v in ts.items()})
    print(f"  converged N={N} Re={Re:g}: {s.metrics.iterations} its, {time.time() - t0:.1f}s")


def g7b_converged_n64():
    """BASELINE config 2: solver=spectral N=64 Re=400 to the reference's stopping rule (273 012 iterations,
    about 6-7 minutes of CPU here)."""
    g7_converged(64, 400.0)


def g11_interp():
    """spectral_interpolate (the Ghia-centreline recipe, polynomial.py:398) on CGL nodes."""
    _install_shims()
    poly = importlib.import_module("solvers.spectral.basis.polynomial")
    rng = np.random.default_rng(3)
    out = {}
    for N in (16, 32, 64):
        x = 0.5 * (1.0 - np.cos(np.pi * np.arange(N + 1) / N))
        f = np.sin(3 * x) + 0.3 * rng.standard_normal(N + 1) * 1e-3
        xe = np.linspace(0.0, 1.0, 17)
        out[f"N{N}_x"] = x
        out[f"N{N}_f"] = f
        out[f"N{N}_xe"] = xe
        out[f"N{N}_fe"] = poly.spectral_interpolate(x, f, xe, basis="legendre")
    np.savez_compressed(OUT / "g11_interp.npz", **out)


def g12_legendre():
    """basis_type='legendre' (sg.py:56-59): operators and short trajectories."""
    out = {}
    for N in (8, 16, 33):
        s = make_sg(N, 100.0, basis_type="legendre")
        out[f"N{N}_x"] = s.basis_x.nodes(N + 1)
        out[f"N{N}_Dx"] = s.Dx_1d
        out[f"N{N}_Dxx"] = s.Dxx_1d
        out[f"N{N}_Interp_x"] = s.Interp_x
        out[f"N{N}_w_x"] = s.w_x
        out[f"N{N}_dx_min"] = np.array(s.dx_min)
    for N, Re, K in ((16, 100.0, 60), (32, 400.0, 300)):
        s = make_sg(N, Re, basis_type="legendre")
        h = _run_steps(s, K)
        tag = f"T{N}"
        out[f"{tag}_u"], out[f"{tag}_v"], out[f"{tag}_p"] = s.arrays.u.copy(), s.arrays.v.copy(), s.arrays.p.copy()
        for k, val in h.items():
            out[f"{tag}_{k}"] = val
    np.savez_compressed(OUT / "g12_legendre.npz", **out)
    (OUT / "g12_legendre.json").write_text(json.dumps({"T16": dict(N=16, Re=100.0, K=60),
                                                       "T32": dict(N=32, Re=400.0, K=300)}, indent=1))


def make_fsg(N, Re, **kw):
    _install_shims()
    fsg = importlib.import_module("solvers.spectral.fsg")
    args = dict(
        name="spectral_fsg", Re=Re, lid_velocity=1.0, Lx=1.0, Ly=1.0, nx=N, ny=N,
        tolerance=1e-6, max_iterations=500000, basis_type="chebyshev", CFL=1.5,
        beta_squared=5.0, corner_treatment="smoothing", corner_smoothing=0.15,
        multigrid="fsg", n_levels=2, coarse_tolerance_factor=1.0,
        prolongation_method="fft", restriction_method="fft",
    )
    args.update(kw)
    with kron_guard():
        return fsg.FSGSolver(**args)


def g8_fsg(full=False):
    """FSG path (a20): prolongation operator (quirk Q10), capped two-level runs (quirk Q2), the
    single-level fallback of build_hierarchy, and one fully converged FSG solve."""
    _install_shims()
    tr = importlib.import_module("solvers.spectral.operators.transfer_operators")
    rng = np.random.default_rng(8)
    out = {}
    pro = tr.FFTProlongation()
    for nc, nf in ((17, 33), (15, 31), (13, 25), (9, 19)):
        f = rng.standard_normal((nc, nc))
        out[f"pro_in_{nc}_{nf}"] = f
        out[f"pro_out_{nc}_{nf}"] = pro.prolongate_2d(f, (nf, nf))
        out[f"pro_mat_{nc}_{nf}"] = np.stack([pro.prolongate_1d(e, nf) for e in np.eye(nc)], axis=1)
    poly = tr.PolynomialProlongation()                  # prolongation_method="polynomial" (transfer_operators.py:333-376)
    for nc, nf in ((17, 33), (15, 31), (9, 19)):
        out[f"poly_mat_{nc}_{nf}"] = np.stack([poly.prolongate_1d(e, nf) for e in np.eye(nc)], axis=1)
        out[f"poly_out_{nc}_{nf}"] = poly.prolongate_2d(out[f"pro_in_{nc}_{nf}"], (nf, nf))
    np.savez_compressed(OUT / "g8_prolongation.npz", **out)

    runs = {}
    cases = [("cap300_N32_Re100", 32, 100.0, dict(max_iterations=300)),
             ("cap200_N24_Re400", 24, 400.0, dict(max_iterations=200, corner_smoothing=0.1)),
             ("single_N20_Re100", 20, 100.0, dict(max_iterations=250)),
             ("cap150_N48_Re1000_saad", 48, 1000.0, dict(max_iterations=150, corner_treatment="saad")),
             ("lvl3_N48_Re100", 48, 100.0, dict(max_iterations=120, n_levels=3, coarse_tolerance_factor=10.0)),
             # round 3: a two-level run whose coarse level is N=32 and fine level N=64 (tail layout on both), and the
             # reference's other prolongation (create_transfer_operators, transfer_operators.py:526-527)
             ("cap100_N64_Re1000", 64, 1000.0, dict(max_iterations=100)),
             ("poly_cap200_N32_Re400", 32, 400.0, dict(max_iterations=200, prolongation_method="polynomial"))]
    if full:
        cases.append(("full_N32_Re100", 32, 100.0, dict()))
    meta = {}
    # entries this call does not regenerate (the converged run without --full) are carried over from the committed files
    if (OUT / "g8_fsg_runs.npz").exists():
        names = {c[0] for c in cases}
        old_meta = json.loads((OUT / "g8_fsg_runs.json").read_text())
        with np.load(OUT / "g8_fsg_runs.npz") as old:
            for key in old.files:
                if key.rsplit("_", 1)[0] not in names:
                    runs[key] = old[key]
        meta.update({k: v for k, v in old_meta.items() if k not in names})
    for name, N, Re, kw in cases:
        t0 = time.time()
        with kron_guard():
            s = make_fsg(N, Re, **kw)
            s.solve()
        runs[f"{name}_u"], runs[f"{name}_v"], runs[f"{name}_p"] = s.arrays.u.copy(), s.arrays.v.copy(), s.arrays.p.copy()
        m = {k: (v.item() if isinstance(v, np.generic) else v) for k, v in s.metrics.__dict__.items()}
        ts = {k: (list(map(float, v)) if v else []) for k, v in s.time_series.__dict__.items()}
        meta[name] = dict(N=N, Re=Re, kw=kw, metrics=m, time_series=ts)
        print(f"  fsg {name}: {m['iterations']} its converged={m['converged']} {time.time() - t0:.1f}s")
    np.savez_compressed(OUT / "g8_fsg_runs.npz", **runs)
    (OUT / "g8_fsg_runs.json").write_text(json.dumps(meta, indent=1))


GROUPS = {
    "G1": g1_operators, "G2": g2_lid, "G3": g3_single_stage, "G4": g4_trajectories,
    "G4b": g4b_variants, "G4c": g4c_headline_sizes, "G7": g7_converged, "G7b": g7b_converged_n64, "G11": g11_interp, "G8": g8_fsg, "G12": g12_legendre, "G13": g13_unequal_grids,
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--full", action="store_true", help="also the 3000-step N=64 Re=1000 trajectory")
    a = ap.parse_args()
    todo = [g.strip() for g in a.only.split(",") if g.strip()] or [g for g in GROUPS if g != "G7b" or a.full]
    for g in todo:
        t0 = time.time()
        print(f"[{g}]")
        if g in ("G4", "G8"):
            GROUPS[g](full=a.full)
        else:
            GROUPS[g]()
        print(f"[{g}] done in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
