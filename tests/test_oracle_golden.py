"""Pin the CPU oracle (oracle/ldc_oracle.py) to golden vectors produced by running the
reference itself (tests/golden/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest

from oracle import ldc_oracle as orc


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(golden_dir / "g1_operators.npz")


@pytest.mark.parametrize("N", [8, 16, 33, 64])
def test_operators(g1, N):
    ax = orc.Axis(N, 1.0)
    assert rel(ax.x, g1[f"N{N}_x"]) == 0.0
    assert rel(ax.D, g1[f"N{N}_Dx"]) < 1e-13
    assert rel(ax.D2, g1[f"N{N}_Dxx"]) < 1e-12
    assert rel(ax.I, g1[f"N{N}_Interp_x"]) < 1e-13
    assert rel(ax.w, g1[f"N{N}_w_x"]) < 1e-13
    assert ax.hmin == float(g1[f"N{N}_dx_min"])


def test_operators_scaled_domain(g1):
    ax, ay = orc.Axis(12, 2.0), orc.Axis(12, 0.5)
    assert rel(ax.x, g1["L_x"]) < 1e-15 and rel(ay.x, g1["L_y"]) < 1e-15
    assert rel(ax.D, g1["L_Dx"]) < 1e-13 and rel(ay.D, g1["L_Dy"]) < 1e-13
    assert rel(ax.w, g1["L_w_x"]) < 1e-13 and rel(ay.w, g1["L_w_y"]) < 1e-13
    assert rel(ay.I, g1["L_Interp_y"]) < 1e-13


def test_lid_profiles(golden_dir):
    g = np.load(golden_dir / "g2_lid.npz")
    x = orc.Axis(32, 1.0).x
    for cs in (0.0, 0.01, 0.15, 0.35, 0.5):
        got = orc.lid_profile(x, "smoothing", cs, 1.0, 1.0)
        assert np.max(np.abs(got - g[f"smooth_{cs}"])) <= 2.3e-16
    assert np.max(np.abs(orc.lid_profile(x, "saad", 0.15, 1.0, 1.0) - g["saad"])) <= 2.3e-16
    assert np.max(np.abs(orc.lid_profile(x, "polynomial", 0.15, 1.0, 1.0) - g["saad"])) <= 2.3e-16
    x2 = orc.Axis(20, 2.0).x
    got = orc.lid_profile(x2, "smoothing", 0.2, 2.5, 2.0)
    assert np.max(np.abs(got - g["smooth_L2_U2.5"])) <= 6e-16
    with pytest.raises(ValueError):
        orc.lid_profile(x, "subtraction", 0.1, 1.0, 1.0)


@pytest.mark.parametrize("N", [16, 24])
def test_single_residual(golden_dir, N):
    g = np.load(golden_dir / "g3_single_stage.npz")
    M = N + 1
    s = orc.OracleSG(N, 400.0)
    s.u = g[f"N{N}_u"].reshape(M, M).copy()
    s.v = g[f"N{N}_v"].reshape(M, M).copy()
    s.p = g[f"N{N}_p"].reshape(M - 2, M - 2).copy()
    Ru, Rv, Rp, parts = s.residual(s.u, s.v, s.p, want_parts=True)
    for k, val in parts.items():
        assert rel(val.ravel(), g[f"N{N}_{k}"]) < 1e-12, k
    assert rel(Ru.ravel(), g[f"N{N}_R_u"]) < 1e-12
    assert rel(Rv.ravel(), g[f"N{N}_R_v"]) < 1e-12
    assert rel(Rp.ravel(), g[f"N{N}_R_p"]) < 1e-12


def _run(s, K):
    out = dict(dt=[], rel=[], res=[], E=[], Z=[], P=[])
    up, vp = s.u.copy(), s.v.copy()
    for _ in range(K):
        out["dt"].append(s.step())
        du = np.linalg.norm(s.u - up) / (np.linalg.norm(up) + 1e-12)
        dv = np.linalg.norm(s.v - vp) / (np.linalg.norm(vp) + 1e-12)
        out["rel"].append(max(du, dv))
        out["res"].append(s.residual_norms())
        out["E"].append(s.energy()); out["Z"].append(s.enstrophy()); out["P"].append(s.palinstrophy())
        up, vp = s.u.copy(), s.v.copy()
    return {k: np.array(v) for k, v in out.items()}


TRAJ = [(16, 100, 50), (32, 100, 500), (64, 400, 1000)]


@pytest.mark.parametrize("N,Re,K", TRAJ)
def test_trajectory(golden_dir, N, Re, K):
    """K steps from rest: state <= 1e-12 abs, dt/rel/E/Z/P/residual norms <= 1e-11 rel."""
    g = np.load(golden_dir / f"g4_traj_N{N}_Re{Re}_K{K}.npz")
    s = orc.OracleSG(N, float(Re))
    h = _run(s, K)
    assert np.max(np.abs(s.u.ravel() - g["u"])) < 1e-12
    assert np.max(np.abs(s.v.ravel() - g["v"])) < 1e-12
    assert np.max(np.abs(s.p.ravel() - g["p"])) < 1e-12
    assert rel(h["dt"], g["dt"]) < 1e-13
    # the first step starts from v = 0, p = 0: compare with an absolute floor
    assert np.max(np.abs(h["rel"] - g["rel"]) / (np.abs(g["rel"]) + 1e-9)) < 1e-9
    assert rel(h["res"], g["res"]) < 1e-11
    for k in ("E", "Z", "P"):
        assert rel(h[k], g[k]) < 1e-11, k
    assert rel(s.vorticity().ravel(), g["omega"]) < 1e-11
    # psi and the vortex table (reference: sparse LU of the Kronecker system)
    psi = s.streamfunction()
    assert np.max(np.abs(psi - g["psi"])) < 1e-11 * max(np.max(np.abs(g["psi"])), 1e-30) + 1e-15
    vm = s.vortex_metrics(psi)
    for key, ref in zip(g["vortex_keys"], g["vortex_vals"]):
        assert abs(vm[str(key)] - ref) <= 1e-10 * max(abs(ref), 1.0), key


def compact_diffs(a, g, name):
    """Largest deviations of a 2-D array from the compact form a G4c fixture holds (make_golden._compact_state)."""
    rows, cols = g[f"{name}_rows_idx"], g[f"{name}_cols_idx"]
    d = max(np.max(np.abs(a[rows, :] - g[f"{name}_rows"])), np.max(np.abs(a[:, cols] - g[f"{name}_cols"])),
            np.max(np.abs(a[::8, ::8] - g[f"{name}_lattice"])))
    dn = abs(np.linalg.norm(a) - float(g[f"{name}_norm2"])) / max(float(g[f"{name}_norm2"]), 1e-300)
    dm = abs(np.max(np.abs(a)) - float(g[f"{name}_max"]))
    return d, dn, dm


G4C = [(128, 1000, 40), (256, 1000, 25), (128, 1000, 400), (256, 1000, 200)]


@pytest.mark.parametrize("N,Re,K", G4C)
def test_trajectory_at_headline_sizes(golden_dir, N, Re, K):
    """The reference itself at N=128 / N=256, Re=1000 (BASELINE configs 3-5; sg.py:410-449, base.py:250-276): every
    record in full, the end state on six rows, six columns and the every-8th-node lattice, its norm and maximum."""
    g = np.load(golden_dir / f"g4c_traj_N{N}_Re{Re}_K{K}.npz")
    s = orc.OracleSG(N, float(Re))
    h = _run(s, K)
    for name, a in (("u", s.u), ("v", s.v), ("p", s.p)):
        d, dn, dm = compact_diffs(a, g, name)
        assert d < 1e-12 and dn < 1e-12 and dm < 1e-12, (name, d, dn, dm)
    d, dn, dm = compact_diffs(s.vorticity(), g, "omega")
    assert d < 1e-11 * float(g["omega_max"]) and dn < 1e-11, (d, dn, dm)
    assert rel(h["dt"], g["dt"]) < 1e-13
    assert np.max(np.abs(h["rel"] - g["rel"]) / (np.abs(g["rel"]) + 1e-9)) < 1e-9
    assert rel(h["res"], g["res"]) < 1e-11
    for k in ("E", "Z", "P"):
        assert rel(h[k], g[k]) < 1e-11, k


@pytest.mark.parametrize("N", [8, 16, 33])
def test_legendre_operators(golden_dir, N):
    """basis_type='legendre' (sg.py:56-59): LGL nodes, D = Vx V^-1, LGL weights."""
    g = np.load(golden_dir / "g12_legendre.npz")
    ax = orc.Axis(N, 1.0, "legendre")
    assert rel(ax.x, g[f"N{N}_x"]) < 1e-15
    assert rel(ax.D, g[f"N{N}_Dx"]) < 1e-12
    assert rel(ax.D2, g[f"N{N}_Dxx"]) < 1e-11
    assert rel(ax.I, g[f"N{N}_Interp_x"]) < 1e-12
    assert rel(ax.w, g[f"N{N}_w_x"]) < 1e-13
    assert abs(ax.hmin - float(g[f"N{N}_dx_min"])) < 1e-15


@pytest.mark.parametrize("tag", ["T16", "T32"])
def test_legendre_trajectory(golden_dir, tag):
    g = np.load(golden_dir / "g12_legendre.npz")
    c = json.loads((golden_dir / "g12_legendre.json").read_text())[tag]
    s = orc.OracleSG(c["N"], c["Re"], basis_type="legendre")
    h = _run(s, c["K"])
    for k, a in (("u", s.u), ("v", s.v), ("p", s.p)):
        assert np.max(np.abs(a.ravel() - g[f"{tag}_{k}"])) < 1e-11, k
    assert rel(h["dt"], g[f"{tag}_dt"]) < 1e-12
    assert rel(h["res"], g[f"{tag}_res"]) < 1e-10
    for k in ("E", "Z", "P"):
        assert rel(h[k], g[f"{tag}_{k}"]) < 1e-10, k


def test_variants(golden_dir):
    g = np.load(golden_dir / "g4b_variants.npz")
    meta = json.loads((golden_dir / "g4b_variants.json").read_text())
    for name, c in meta.items():
        s = orc.OracleSG(c["N"], c["Re"], **c["kw"])
        h = _run(s, c["K"])
        assert np.max(np.abs(s.u.ravel() - g[f"{name}_u"])) < 1e-12, name
        assert np.max(np.abs(s.v.ravel() - g[f"{name}_v"])) < 1e-12, name
        assert np.max(np.abs(s.p.ravel() - g[f"{name}_p"])) < 1e-12, name
        assert rel(h["dt"], g[f"{name}_dt"]) < 1e-13, name
        assert rel(h["P"], g[f"{name}_P"]) < 1e-10, name


def test_q1_stage_pressure_flag_changes_trajectory():
    """Quirk Q1: SG differentiates p^n in all four stages; the smoother variant does not."""
    a, b = orc.OracleSG(16, 100.0), orc.OracleSG(16, 100.0, stage_pressure=True)
    for _ in range(10):
        a.step(); b.step()
    assert np.max(np.abs(a.u - b.u)) > 1e-5


def test_spectral_interpolate(golden_dir):
    g = np.load(golden_dir / "g11_interp.npz")
    for N in (16, 32, 64):
        got = orc.spectral_interpolate(g[f"N{N}_x"], g[f"N{N}_f"], g[f"N{N}_xe"])
        assert np.max(np.abs(got - g[f"N{N}_fe"])) < 1e-11


def test_converged_run_matches_reference(golden_dir):
    """Full solve at N=32, Re=100 with the reference's stopping rule (59 649 iterations)."""
    meta = json.loads((golden_dir / "g7_converged_N32_Re100.json").read_text())
    g = np.load(golden_dir / "g7_converged_N32_Re100.npz")
    s = orc.OracleSG(32, 100.0)
    its, conv, hist = s.solve(tolerance=1e-6)
    m = meta["metrics"]
    assert conv and m["converged"]
    assert abs(its - m["iterations"]) <= 2
    if its == m["iterations"]:
        assert np.max(np.abs(s.u.ravel() - g["u"])) < 1e-11
        assert np.max(np.abs(s.pressure_on_full_grid().ravel() - g["p"])) < 1e-10
        assert abs(hist["E"][-1] - m["final_energy"]) < 1e-11
        assert abs(hist["P"][-1] - m["final_palinstrophy"]) < 1e-8 * m["final_palinstrophy"]
    vm = s.vortex_metrics()
    assert abs(vm["psi_min"] - m["psi_min"]) < 1e-9
    assert vm["psi_min_x"] == pytest.approx(m["psi_min_x"], abs=1e-12)
    assert vm["psi_min_y"] == pytest.approx(m["psi_min_y"], abs=1e-12)
