"""CPU-only checks of the boundary: the C-ABI library loads and exports every symbol the
header declares, argument validation works without a device, and the plugin class refuses
to run without a GPU (no silent fallback)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from solvers.spectral import ldc_lib
    return ldc_lib


def test_header_and_exports_agree(lib):
    hdr = (ROOT / "include" / "ldc_hip.h").read_text()
    declared = set(re.findall(r"\b(ldc_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(lib.EXPORTS)
    L = lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.ldc_version() == lib.ABI_VERSION == 7


def test_product_library_has_no_timing_switches(lib):
    """The timing switches (ldc_debug_ablate / ldc_debug_stamps) are compiled out of the product library; the
    instrumented build (-DLDC_TIMING, tools/ only) is a file of its own and says so."""
    L = lib.lib()
    assert L.ldc_timing_build() == 0
    fake = C.c_void_p(1)                                  # never dereferenced by the product build
    assert L.ldc_debug_ablate(fake, 3) == -2 and L.ldc_debug_stamps(fake, None) == -2      # LDC_E_STATE
    assert lib.TIMING_LIB_PATH.exists() and lib.TIMING_LIB_PATH != lib.LIB_PATH
    T = C.CDLL(str(lib.TIMING_LIB_PATH))
    assert T.ldc_timing_build() == 1 and T.ldc_version() == lib.ABI_VERSION
    assert T.ldc_debug_ablate(None, 0) == -2              # a null handle is still refused there


def test_python_constants_match_the_header(lib):
    hdr = (ROOT / "include" / "ldc_hip.h").read_text()
    val = lambda name: int(re.search(rf"{name}\s*=?\s*(-?\d+)", hdr).group(1))        # noqa: E731
    assert val("LDC_SYNC_GIVEUP") == lib.SYNC_GIVEUP and val("LDC_SYNC_LEN") == lib.SYNC_LEN
    assert val("#define LDC_XCD_TILES") == lib.XCD_TILES and val("#define LDC_XCD_AUTO_TILES") == lib.XCD_AUTO_TILES
    assert val("#define LDC_CU_MAX_M") == lib.CU_MAX_M and val("#define LDC_CU_AUTO_TRIALS ") == lib.CU_AUTO_TRIALS
    assert val("#define LDC_CU_AUTO_TRIALS_T3") == lib.CU_AUTO_TRIALS_T3
    assert val("#define LDC_CU_AUTO_TRIALS_M33") == lib.CU_AUTO_TRIALS_M33
    assert val("#define LDC_NPART") == lib.NPART and val("#define LDC_ABI_VERSION") == lib.ABI_VERSION


def test_struct_size_matches_header(lib):
    # 4 int32 + 7 double + 4 int32 + (37 + 27 packed twins) ptr + int64 + 4 ptr (scal, ctrl, rec, sync) + 2 int32 (Mx, My)
    assert C.sizeof(lib.Problem) == 16 + 56 + 16 + (37 + 27) * 8 + 8 + 32 + 8


def test_argument_validation_needs_no_device(lib):
    L = lib.lib()
    h = C.c_void_p()
    pr = lib.Problem()
    assert L.ldc_solver_create(C.byref(pr), C.byref(h)) == -1          # LDC_E_ARG
    assert L.ldc_solver_create(None, C.byref(h)) == -1
    assert L.ldc_stage(None, 0, None) == -2                             # LDC_E_STATE
    assert L.ldc_solver_enqueue(None, 1, 1, None) == -2
    assert L.ldc_solver_set_persistent(None, 1) == -2 and L.ldc_solver_status(None) == -2
    assert b"barrier" in L.ldc_error_string(-4)                          # LDC_E_SYNC
    assert L.ldc_gemm_nt(None, None, None, 1, 1, 16, 0, 0, None, None, None) == -1
    assert b"invalid argument" in L.ldc_error_string(-1)


def test_geometry_rules(lib):
    """T = ceil((M-1)/16); tail iff 16T == M-1; LD multiple of 16 and >= 16T+16."""
    L = lib.lib()
    pr = lib.Problem()
    pr.M, pr.T, pr.tail, pr.LD, pr.rec_cap = 257, 16, 1, 272, 8
    h = C.c_void_p()
    # geometry is consistent but pointers are null -> still E_ARG, never a crash
    assert L.ldc_solver_create(C.byref(pr), C.byref(h)) == -1
    pr.T = 17
    assert L.ldc_solver_create(C.byref(pr), C.byref(h)) == -1


def test_plugin_refuses_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from solvers.spectral.sg import SGSolver
    with pytest.raises(lib.LdcError, match="no CPU fallback"):
        SGSolver(name="spectral", Re=100.0, nx=16, ny=16, basis_type="chebyshev", CFL=1.5)


def test_constructor_errors_match_reference():
    from solvers.spectral.sg import SGSolver
    with pytest.raises(TypeError):
        SGSolver(name="spectral", Re=100.0, nx=16, ny=16, bogus_key=1)
    with pytest.raises(ValueError, match="Unknown basis_type"):
        SGSolver(name="spectral", Re=100.0, nx=16, ny=16, basis_type="fourier")
    with pytest.raises(ValueError, match="Unknown corner treatment"):
        SGSolver(name="spectral", Re=100.0, nx=16, ny=16, basis_type="chebyshev",
                 corner_treatment="subtraction")
