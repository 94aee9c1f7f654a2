"""bench.py's host-side helpers (no GPU): the hipGraph length it captures for a given --steps, the flop count of
SURVEY 8d it prices the roofline with, and the fields of the committed bench line of the round."""
import importlib.util
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_graph_length_divides_the_timed_steps():
    b = _bench()
    for K in (1, 5, 20, 32, 50, 64, 100, 512, 2000, 4000):
        g = b.graph_iters_for(K)
        assert 1 <= g <= 64
        if K <= 64:
            assert g == K                       # one capture holds the whole region
        else:
            assert K % g == 0 or g == 64        # whole replays, or 64 with an eager remainder
    assert b.graph_iters_for(4000) == 50 and b.graph_iters_for(20) == 20 and b.graph_iters_for(67) == 64


def test_flop_count_is_the_necessary_one():
    b = _bench()
    M, Mi = 257, 255
    assert b.flops_per_step(256, False) == 68.0 * M**3 + 2.0 * M * Mi * (M + Mi)
    assert b.flops_per_step(256, True) - b.flops_per_step(256, False) == 8.0 * M**3


def test_ghia_block_carries_the_metrics_second_half():
    """BASELINE's metric is "time-steps/s ...; Ghia centreline L2 error": the bench line carries both figures of the
    bench config (reference stopping rule; converged to 1e-9) from the committed reports, labelled as such."""
    b = _bench()
    g = b.ghia_block(256, 1000.0)
    assert "not a live measurement" in g["kind"]
    ref, tight = g["reference_stopping_rule_tol_1e-6"], g["converged_tol_1e-9"]
    assert ref["iterations"] == 1299041 and abs(ref["u_rms"] - 0.2197) < 1e-3 and abs(ref["v_rms"] - 0.2901) < 1e-3
    assert tight["iterations"] == 64603605 and abs(tight["u_rms"] - 0.0103) < 1e-4 and abs(tight["v_rms"] - 0.0071) < 1e-4
    assert (ROOT / ref["source"]).exists() and (ROOT / tight["source"]).exists()
    assert b.ghia_block(48, 1000.0) is None                      # no committed solve for that size


def test_committed_bench_line_keeps_the_contract():
    files = sorted((ROOT / "profiles").glob("r*_bench.json"))
    assert files, "no committed bench line"
    d = json.loads(files[-1].read_text())
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["dtype"] == "f64" and d["unit"] == "steps/s" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and "sample" in c
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6


def test_one_batch_runs_in_the_callers_thread_without_a_gpu():
    """run_concurrently with a single item is a plain call (LDC_BATCH_STREAMS=1, or a rank with one batch): no stream,
    no thread, exceptions pass straight through."""
    import sys
    import threading
    sys.path.insert(0, str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src"))
    from solvers.spectral.batched import run_concurrently
    seen = []
    wall = run_concurrently(["only"], lambda b: seen.append((b, threading.current_thread() is threading.main_thread())))
    assert seen == [("only", True)] and wall >= 0.0
    import pytest
    with pytest.raises(ZeroDivisionError):
        run_concurrently(["x"], lambda b: 1 / 0)


def test_plain_gpus_2_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2 ...` started PLAINLY (no WORLD_SIZE): the process becomes the launcher -- two ranks of itself
    through torch.distributed.run before anything touches a GPU --, relays rank 0's one JSON line and the ranks' exit code
    (reference model of the fan-out: scripts/hpc_submit.py:103-107, 182-200).  --dry-run keeps the GPU out of it: rendezvous on
    127.0.0.1, barrier, max over ranks, gather, all over gloo on CPU."""
    import os
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["LDC_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--dry-run"],
                       capture_output=True, text=True, env=env, timeout=600, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["ranks_seen"] == [0, 1] and d["dry_run"] is True
    assert abs(d["max_over_ranks_s"] - 2e-3) < 1e-12
    # a rank that fails takes the launcher's exit code with it (--gpus 3 inside a 2-rank world is refused by every rank)
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r2 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "3", "--dry-run"], capture_output=True, text=True,
                        env=env2, timeout=120, cwd=tmp_path)
    assert r2.returncode != 0 and "WORLD_SIZE=2" in (r2.stderr + r2.stdout)
