"""Host-side setup code of the product (NumPy) against the reference's golden vectors and
against the oracle.  CPU only; no device call."""
import numpy as np
import pytest

from oracle import ldc_oracle as orc
from solvers.spectral.basis.spectral import (ChebyshevLobattoBasis, LegendreLobattoBasis, clenshaw_curtis_weights,
                                              inner_to_full_interpolation)
from solvers.spectral.basis.polynomial import spectral_interpolate
from solvers.spectral.operators.corner import create_corner_treatment
from solvers.datastructures import Metrics, SpectralParameters, TimeSeries


def rel(a, b):
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("N", [8, 16, 33, 64])
def test_basis_matches_reference(golden_dir, N):
    g = np.load(golden_dir / "g1_operators.npz")
    b = ChebyshevLobattoBasis(domain=(0.0, 1.0))
    x = b.nodes(N + 1)
    D = b.diff_matrix(x)
    assert rel(x, g[f"N{N}_x"]) == 0.0
    assert rel(D, g[f"N{N}_Dx"]) < 1e-13
    assert rel(D @ D, g[f"N{N}_Dxx"]) < 1e-12
    assert rel(inner_to_full_interpolation(x[1:-1], x), g[f"N{N}_Interp_x"]) < 1e-13
    assert rel(b.quadrature_weights(N + 1), g[f"N{N}_w_x"]) < 1e-13
    assert abs(clenshaw_curtis_weights(N + 1).sum() - 2.0) < 1e-13


@pytest.mark.parametrize("N", [8, 16, 33])
def test_legendre_basis_matches_reference(golden_dir, N):
    g = np.load(golden_dir / "g12_legendre.npz")
    b = LegendreLobattoBasis(domain=(0.0, 1.0))
    x = b.nodes(N + 1)
    D = b.diff_matrix(x)
    assert rel(x, g[f"N{N}_x"]) < 1e-15
    assert rel(D, g[f"N{N}_Dx"]) < 1e-12
    assert rel(D @ D, g[f"N{N}_Dxx"]) < 1e-11
    assert rel(inner_to_full_interpolation(x[1:-1], x), g[f"N{N}_Interp_x"]) < 1e-12
    assert rel(b.quadrature_weights(N + 1), g[f"N{N}_w_x"]) < 1e-13
    assert abs(b.quadrature_weights(N + 1).sum() - 1.0) < 1e-13


def test_lid_profiles_match_reference(golden_dir):
    g = np.load(golden_dir / "g2_lid.npz")
    x = ChebyshevLobattoBasis(domain=(0.0, 1.0)).nodes(33)
    for cs in (0.0, 0.01, 0.15, 0.35, 0.5):
        u, v = create_corner_treatment("smoothing", cs).get_lid_velocity(x, x * 0 + 1, 1.0, 1.0, 1.0)
        assert np.max(np.abs(u - g[f"smooth_{cs}"])) <= 2.3e-16 and not v.any()
    u, _ = create_corner_treatment("saad").get_lid_velocity(x, x, 1.0, 1.0, 1.0)
    assert np.max(np.abs(u - g["saad"])) <= 2.3e-16
    uw, vw = create_corner_treatment("smoothing").get_wall_velocity(x, x, 1.0, 1.0)
    assert not uw.any() and not vw.any()


def test_spectral_interpolate_matches_reference(golden_dir):
    g = np.load(golden_dir / "g11_interp.npz")
    for N in (16, 32, 64):
        got = spectral_interpolate(g[f"N{N}_x"], g[f"N{N}_f"], g[f"N{N}_xe"])
        assert np.max(np.abs(got - g[f"N{N}_fe"])) < 1e-11
    with pytest.raises(ValueError):
        spectral_interpolate(g["N16_x"], g["N16_f"], g["N16_xe"], basis="fourier")


def test_parameters_and_metrics_surface():
    p = SpectralParameters(name="spectral", Re=400, nx=64, ny=64, CFL=1.5, basis_type="chebyshev")
    d = p.to_mlflow()
    for k in ("name", "Re", "lid_velocity", "Lx", "Ly", "nx", "ny", "max_iterations", "tolerance", "method",
              "basis_type", "CFL", "beta_squared", "corner_treatment", "corner_smoothing", "multigrid",
              "n_levels", "coarse_tolerance_factor", "prolongation_method", "restriction_method"):
        assert k in d
    assert d["method"] == "Spectral-AC" and "device" not in d
    m = Metrics()
    out = m.to_mlflow()
    assert "final_residual" not in out            # +inf is dropped
    assert out["converged"] == 0 and isinstance(out["converged"], int)
    assert len(out) == 28
    ts = TimeSeries(rel_iter_residual=[1.0, 0.5], energy=[0.1, None])
    assert ("energy", 0, 0.1) in ts.to_records() and len(ts.to_records()) == 3
