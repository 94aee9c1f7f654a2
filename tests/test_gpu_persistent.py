"""Modes 1 and 2 of ldc_solver_set_persistent (the round-2 persistent trial kernel) are not part of the product
library: csrc/ldc_trial_kernel.inc is compiled into the instrumented build only (it lost to the launch path at
every size and to the small-N kernel, mode 3, where a persistent kernel pays -- profiles/r02_persist_*.log,
profiles/r03_xcd_ab.log).  What the product must do with them: refuse at the C-ABI, and run the launch path when a
configuration still asks for them.  (Mode 3 has its own file: tests/test_gpu_xcd.py.)"""
import numpy as np
import pytest

from test_gpu_parity import make

pytestmark = pytest.mark.gpu


def test_product_library_refuses_the_round2_persistent_modes():
    from solvers.spectral import ldc_lib as L
    assert L.lib().ldc_timing_build() == 0
    s = make(32, 100.0, persistent=0)
    s._begin(0.0)
    assert L.lib().ldc_solver_set_persistent(s._handle, 7) == -1          # LDC_E_ARG
    assert L.lib().ldc_solver_set_persistent(s._handle, 1) == -1
    assert L.lib().ldc_solver_set_persistent(s._handle, 2) == -1
    assert L.lib().ldc_solver_set_persistent(s._handle, 3) == 0           # 3 x 3 work-groups: fits one XCD
    assert L.lib().ldc_solver_mode(s._handle) == 3
    assert L.lib().ldc_solver_set_persistent(s._handle, 0) == 0
    assert L.lib().ldc_solver_mode(s._handle) == 0
    s.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_configurations_asking_for_them_run_the_launch_path(mode):
    from solvers.spectral import ldc_lib as L
    a = make(48, 100.0, persistent=mode, check_every=256)
    ra = a.run_iterations(300, diagnostics=True)
    assert L.lib().ldc_solver_mode(a._handle) == 0
    b = make(48, 100.0, persistent=0, check_every=256)
    rb = b.run_iterations(300, diagnostics=True)
    assert np.array_equal(ra, rb)
    for x, y in ((a.arrays.u, b.arrays.u), (a.arrays.v, b.arrays.v), (a.arrays.p, b.arrays.p)):
        assert np.array_equal(x, y)
    a.close(); b.close()
