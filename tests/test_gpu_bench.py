"""bench.py on the GPU box through the PLAIN command line the driver uses."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _run(args, extra_env=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_plain_gpus_2_on_one_card():
    """`python bench.py --gpus 2 ...` as the contract line says: the process starts its two ranks itself (torch.distributed.run),
    they rendezvous on 127.0.0.1 over gloo, share this box's one card (so the rates mean nothing and the loop is forced onto the
    launch path: two persistent launches cannot both own every CU), rank 0's one line comes back with n_gpus = 2."""
    d = _run(["--gpus", "2", "--steps", "64", "--warmup", "64", "--no-cpu", "--no-farm"], {"LDC_DIST_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["steps"] == 64 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["solver_mode"] == 0 and d["roofline"]["mode"] == 0 and d["config"]["N"] == 256


def test_headline_line_on_one_gpu_takes_the_chip_wide_kernel():
    """The default path of the bench config (N=256, SG with E/Z/P) is the chip-wide kernel; the roofline block prices ITS
    launches (iterations_per_launch x the necessary flops of an iteration) and keeps the launch path's dominant kernel beside."""
    d = _run(["--headline-only", "--steps", "256", "--warmup", "256"])
    r = d["roofline"]
    assert d["n_gpus"] == 1 and d["solver_mode"] == 5 and r["mode"] == 5 and r["bound"] == "mfma"
    assert r["iterations_per_launch"] == 2048 and abs(r["flops_per_launch"] - 2048 * r["flops_per_iteration"]) < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.2 < r["frac"] < 1.0
    assert r["launch_path"]["launch_us"] > 0 and d["value"] > r["launch_path"]["value"] > 0
    assert d["farm"] is None and d["small_n"] is None and d["cu_batch"] is None and d["cpu_baseline"] is None
