#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes of tools/pmc_run.py into one JSON (profiles/rNN_*_pmc.json).

    rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d gpurun_out/pmc_TCC_HIT  -o run --output-format csv -- python3 tools/pmc_run.py
    rocprofv3 --kernel-trace --pmc FETCH_SIZE               -d gpurun_out/pmc_FETCH_SIZE ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE               -d gpurun_out/pmc_WRITE_SIZE ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY \
              SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE            -d gpurun_out/pmc_SQ ...
    python tools/pmc_summarize.py gpurun_out/pmc_TCC_HIT gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_SQ > profiles/r02_pmc.json

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts the wide (128 B) reads at half size, so it is
doubled (MI355X_MICROARCH.md, HBM / rocprofv3 section).  Means are per launch of each library kernel."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KEEP = ("stage_kernel", "post_kernel", "palin_kernel", "finalize_kernel", "prime_kernel", "wide_kernel", "xcd_kernel", "cu_kernel")


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return name.split("(")[0].strip()


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    if not k.startswith(KEEP):
                        continue
                    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    acc[k]["dur_us_" + r["Counter_Name"]].append(
                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, cs in acc.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
        # SQ group (one more pass: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
        # SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE).  SQ_VALU_MFMA_BUSY_CYCLES counts the busy cycles of the matrix pipes
        # summed over the SIMDs: exactly 64 per v_mfma_f64_16x16x4_f64 (the plain stage launch reads 8 388 608 =
        # 131 072 MFMAs x 64).  MFMA utilisation = those cycles over the SIMD-cycles of the launch at the 2.4 GHz peak
        # clock, 1024 SIMDs, with the launch duration of THIS (profiled, hence slower) pass; GRBM_GUI_ACTIVE is kept
        # but not used: its quotient reads high on dispatches this short (MI355X_MICROARCH.md, DVFS give-back).  The
        # wave-level counters are in quad-cycles; their ratios need no unit: WAIT_INST_ANY / WAVE_CYCLES = issue
        # stalls (MFMA dependency / pipe busy), WAIT_ANY / WAVE_CYCLES = waves parked at s_waitcnt / barriers.
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("dur_us_SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
            e["mfma_instructions"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / 64.0
            e["mfma_util"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["dur_us_SQ_VALU_MFMA_BUSY_CYCLES"] * 1e-6 * 2.4e9 * 1024)
        if e.get("SQ_WAVE_CYCLES", 0) > 0:
            for src, dst in (("SQ_WAIT_INST_ANY", "sq_wait_inst_frac"), ("SQ_WAIT_ANY", "sq_wait_any_frac"),
                             ("SQ_ACTIVE_INST_ANY", "sq_active_inst_frac")):
                if src in e:
                    e[dst] = e[src] / e["SQ_WAVE_CYCLES"]
        out[k] = e
    for k, e in out.items():
        if k.startswith(("wide_kernel", "xcd_kernel", "cu_kernel")):
            e["iterations_per_launch"] = 64          # tools/pmc_run.py: the chunk it enqueues (one launch)
    json.dump({"source": "rocprofv3 --kernel-trace --pmc <group> -- python3 tools/pmc_run.py (N=256: one launch-path iteration, "
                         "then ONE launch of the chip-wide kernel with 64 iterations), one group per pass: "
                         "{TCC_HIT_sum,TCC_MISS_sum}, FETCH_SIZE, WRITE_SIZE, the SQ group; "
                         "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts wide reads at half)",
               "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
