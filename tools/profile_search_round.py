"""Host-side profile (cProfile) of ONE search round of BASELINE config 5 through main.py: eight FSG trials at N=128.
    python tools/profile_round.py [n_trials] [n_jobs]        (development aid)"""
import cProfile
import importlib.util
import os
import pstats
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "02689-advancednumericalalgorithmp3_amd"
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g  # noqa: E402

g.build()
spec = importlib.util.spec_from_file_location("ldc_main_prof", PKG / "main.py")
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
n_trials = sys.argv[1] if len(sys.argv) > 1 else "16"
n_jobs = sys.argv[2] if len(sys.argv) > 2 else "8"
work = Path("/tmp/profile_round"); work.mkdir(exist_ok=True)
os.chdir(work)
argv = ["-m", "+experiment/optimization=corner_smoothing", "N=128", f"hydra.sweeper.n_trials={n_trials}",
        f"hydra.sweeper.n_jobs={n_jobs}", "optuna.objective=botella_vortex"]
pr = cProfile.Profile()
pr.enable()
mod.main(argv)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
