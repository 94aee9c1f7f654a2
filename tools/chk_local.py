"""Mode 2 (one-XCD persistent) against the launch path: bit equality of records and state (development aid)."""
import os
import sys
# modes 1 and 2 live in the instrumented build only (csrc/ldc_trial_kernel.inc, -DLDC_TIMING)
os.environ.setdefault("LDC_HIP_LIB", os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "lib", "libldc_hip_timing.so"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import numpy as np
from test_gpu_parity import make

def run(N, K, mode, diag):
    s = make(N, 400.0, persistent=mode, check_every=256)
    rec = s.run_iterations(K, diagnostics=diag)
    out = (rec, s.arrays.u.copy(), s.arrays.v.copy(), s.arrays.p.copy())
    s.close()
    return out

bad = 0
for N in [int(a) for a in sys.argv[1:]] or [16, 24, 32, 48, 64, 80]:
    for diag in (True, False):
        a, b = run(N, 600, 2, diag), run(N, 600, 0, diag)
        same = all(np.array_equal(x, y) for x, y in zip(a, b))
        worst = max(float(np.max(np.abs(x - y))) for x, y in zip(a, b))
        print(f"N={N} diag={int(diag)} identical={same} max|diff|={worst:.3e} finite={bool(np.all(np.isfinite(a[0])))}", flush=True)
        bad += 0 if same else 1
sys.exit(1 if bad else 0)
