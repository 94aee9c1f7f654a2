"""What a trial's record costs on the host after solve(): validation against the FV reference, objective, Ghia error, the
Botella table, results.json, solution.vts (development aid: main.py's make_record, piece by piece).
    python tools/time_record.py [N] [fsg]"""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd", "src"))
import torch  # noqa: E402
from solvers import validation as V  # noqa: E402
from solvers.spectral.sg import SGSolver  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=1e-6, max_iterations=2000,
             check_every=1000, graph_iters=50)
os.chdir(os.path.join(os.getcwd(), "02689-advancednumericalalgorithmp3_amd"))
t0 = time.perf_counter()
s.solve()
torch.cuda.synchronize()
print(f"solve (2000 iterations + _finish): {time.perf_counter() - t0:.3f} s")


def timed(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); best = min(best, time.perf_counter() - t)
    print(f"{label:36s} {best * 1e3:8.1f} ms")
    return out


timed("_finalize_fields", lambda: s._finalize_fields())
timed("compute_vortex_metrics (psi solve)", lambda: s.compute_vortex_metrics())
err = timed("compute_validation_errors", lambda: s.compute_validation_errors(reference_dir="data/validation/fv"))
timed("objective botella_vortex", lambda: V.compute_optuna_objective("botella_vortex", err, s, 1000.0))
timed("validation_table", lambda: s.validation_table())
timed("ghia_error", lambda: s.ghia_error())
out = Path("/tmp/time_record"); out.mkdir(exist_ok=True)
timed("to_vtk().save", lambda: s.to_vtk().save(out / "solution.vts"))
import json  # noqa: E402
timed("params/metrics to_mlflow + json", lambda: (out / "r.json").write_text(json.dumps(dict(p=s.params.to_mlflow(), m=s.metrics.to_mlflow()), default=str)))

# the whole record as main.py makes it, under the profiler (first call of everything, like a sweep's trials)
import cProfile, importlib.util, pstats  # noqa: E402
spec = importlib.util.spec_from_file_location("ldc_main_t", os.path.join(os.getcwd(), "main.py"))
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
s2 = SGSolver(name="spectral", Re=1000.0, nx=N, ny=N, basis_type="chebyshev", CFL=1.5, tolerance=1e-6, max_iterations=500,
              check_every=500, graph_iters=50, corner_smoothing=0.07)
s2.solve()
cfg = dict(solver=dict(name="spectral"), N=N, Re=1000.0, optuna=dict(objective="botella_vortex"), validation=dict(reference_dir="data/validation/fv"))
pr = cProfile.Profile(); t = time.perf_counter(); pr.enable()
mod.make_record(cfg, s2, out / "rec", time.perf_counter())
pr.disable(); print(f"make_record: {(time.perf_counter() - t) * 1e3:.0f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
