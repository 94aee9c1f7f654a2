#!/usr/bin/env python3
"""Which HIP runtime call dies when one host thread creates / destroys streams (and captures graphs on them) while
another thread waits on work of its own?

Round 2 hit an abort inside the HIP runtime when sweeps began to drive the library from two host threads: at the time
`copy_now` created and destroyed a stream per call, every handle created a capture stream of its own and destroyed it
with the handle, and the Python solvers synchronised the whole DEVICE.  The fix (commit 5e75547) serialised all of that
on one never-destroyed stream per device behind a mutex and switched to stream-level waits, but the abort's own output
was not kept.  This probe replays the old call patterns, one variant per child process, against the HIP runtime that
torch loads (the one the product runs under), and records how each child ends.

Variants (thread A | thread B), ITER rounds each:
  create_destroy_vs_stream_sync   A: create stream, small async copy, wait, destroy    | B: kernels on its stream, hipStreamSynchronize
  create_destroy_vs_device_sync   A: the same                                           | B: kernels, hipDeviceSynchronize
  capture_destroy_vs_stream_sync  A: create, capture 8 launches, instantiate, destroy exec + stream | B: kernels, hipStreamSynchronize
  capture_destroy_vs_device_sync  A: the same                                           | B: kernels, hipDeviceSynchronize
  capture_destroy_vs_graph_replay A: the same                                           | B: replays a graph of its own, hipStreamSynchronize
  shared_setup_stream (the fix)   A: copies and captures on ONE kept stream under a lock | B: replays a graph, hipStreamSynchronize

Usage: python tools/probes/stream_race_probe.py [--iters 400] [--out gpurun_out/stream_race.log]
Each child has faulthandler on, so a SIGABRT / SIGSEGV leaves the Python stack of BOTH threads (i.e. which runtime
call each was in) in the log.
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import threading
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
VARIANTS = ["create_destroy_vs_stream_sync", "create_destroy_vs_device_sync", "capture_destroy_vs_stream_sync",
            "capture_destroy_vs_device_sync", "capture_destroy_vs_graph_replay", "shared_setup_stream"]


def child(variant: str, iters: int) -> None:
    import faulthandler
    faulthandler.enable(all_threads=True)
    for p in (str(ROOT), str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src")):
        sys.path.insert(0, p)
    import torch
    from solvers.spectral import ldc_lib as L
    lib = L.lib()
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    vp = C.c_void_p
    hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(vp), C.c_uint]
    hip.hipStreamDestroy.argtypes = [vp]
    hip.hipStreamSynchronize.argtypes = [vp]
    hip.hipMemcpyAsync.argtypes = [vp, vp, C.c_size_t, C.c_int, vp]
    hip.hipStreamBeginCapture.argtypes = [vp, C.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, C.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [C.POINTER(vp), vp, vp, vp, C.c_size_t]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    NONBLOCKING, H2D, RELAXED = 1, 1, 2

    def ok(code, what):
        if code != 0:
            raise RuntimeError(f"{what} -> hipError {code}")

    torch.cuda.init()
    sink_a = torch.zeros(64 * 256, dtype=torch.float64, device="cuda")
    sink_b = torch.zeros(64 * 256, dtype=torch.float64, device="cuda")
    dst = torch.zeros(1024, dtype=torch.float32, device="cuda")
    src = (C.c_float * 1024)(*([1.0] * 1024))
    torch.cuda.synchronize()

    def new_stream():
        s = vp()
        ok(hip.hipStreamCreateWithFlags(C.byref(s), NONBLOCKING), "hipStreamCreateWithFlags")
        return s

    def capture_on(st, sink):
        g, ex = vp(), vp()
        ok(hip.hipStreamBeginCapture(st, RELAXED), "hipStreamBeginCapture")
        for _ in range(8):
            ok(lib.ldc_mfma_peak(sink.data_ptr(), 50, 64, st), "ldc_mfma_peak (captured)")
        ok(hip.hipStreamEndCapture(st, C.byref(g)), "hipStreamEndCapture")
        ok(hip.hipGraphInstantiate(C.byref(ex), g, None, None, 0), "hipGraphInstantiate")
        ok(hip.hipGraphDestroy(g), "hipGraphDestroy")
        return ex

    stop = threading.Event()
    errors = []
    sb = new_stream()
    exec_b = capture_on(sb, sink_b) if variant in ("capture_destroy_vs_graph_replay", "shared_setup_stream") else None
    lock = threading.Lock()
    kept = new_stream() if variant == "shared_setup_stream" else None

    def thread_a():
        try:
            for _ in range(iters):
                if variant == "shared_setup_stream":
                    with lock:
                        ok(hip.hipMemcpyAsync(dst.data_ptr(), C.addressof(src), 4096, H2D, kept), "hipMemcpyAsync")
                        ok(hip.hipStreamSynchronize(kept), "hipStreamSynchronize(kept)")
                        ex = capture_on(kept, sink_a)
                        ok(hip.hipGraphExecDestroy(ex), "hipGraphExecDestroy")
                    continue
                st = new_stream()
                ok(hip.hipMemcpyAsync(dst.data_ptr(), C.addressof(src), 4096, H2D, st), "hipMemcpyAsync")
                ok(hip.hipStreamSynchronize(st), "hipStreamSynchronize(A)")
                if variant.startswith("capture"):
                    ex = capture_on(st, sink_a)
                    ok(hip.hipGraphLaunch(ex, st), "hipGraphLaunch(A)")
                    ok(hip.hipStreamSynchronize(st), "hipStreamSynchronize(A)")
                    ok(hip.hipGraphExecDestroy(ex), "hipGraphExecDestroy")
                ok(hip.hipStreamDestroy(st), "hipStreamDestroy")
        except Exception as exc:
            errors.append(("A", repr(exc)))
        finally:
            stop.set()

    def thread_b():
        try:
            while not stop.is_set():
                if exec_b is not None:
                    ok(hip.hipGraphLaunch(exec_b, sb), "hipGraphLaunch(B)")
                else:
                    ok(lib.ldc_mfma_peak(sink_b.data_ptr(), 50, 64, sb), "ldc_mfma_peak(B)")
                if variant.endswith("device_sync"):
                    ok(hip.hipDeviceSynchronize(), "hipDeviceSynchronize")
                else:
                    ok(hip.hipStreamSynchronize(sb), "hipStreamSynchronize(B)")
        except Exception as exc:
            errors.append(("B", repr(exc)))
            stop.set()

    ta, tb = threading.Thread(target=thread_a, name="A-setup"), threading.Thread(target=thread_b, name="B-wait")
    tb.start(); ta.start()
    ta.join(); tb.join()
    print(f"RESULT {variant}: threads ended, errors={errors}", flush=True)      # before anything that may raise
    try:
        torch.cuda.synchronize()
    except Exception as exc:             # a HIP error left behind surfaces at torch's next check
        print(f"RESULT {variant}: torch.cuda.synchronize() afterwards raised {type(exc).__name__}: {str(exc).splitlines()[0]}", flush=True)
        sys.exit(1)
    sys.exit(1 if errors else 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "stream_race.log"))
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        child(a.child, a.iters)
        return
    out = Path(a.out)
    out.parent.mkdir(parents=True, exist_ok=True)
    with out.open("w") as log:
        for v in VARIANTS:
            try:
                r = subprocess.run([sys.executable, __file__, "--child", v, "--iters", str(a.iters)],
                                   capture_output=True, text=True, timeout=120)
                rc, so, se = r.returncode, r.stdout, r.stderr
            except subprocess.TimeoutExpired as exc:
                rc, so, se = "timeout", (exc.stdout or b"").decode(errors="replace"), (exc.stderr or b"").decode(errors="replace")
            how = f"signal {-rc}" if isinstance(rc, int) and rc < 0 else f"exit {rc}"
            line = f"=== {v}: {how}\n{so[-1500:]}\n--- stderr (tail)\n{se[-3000:]}\n"
            log.write(line); log.flush()
            print(line, flush=True)


if __name__ == "__main__":
    main()
