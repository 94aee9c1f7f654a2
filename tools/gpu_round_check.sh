# the round's evidence run on the GPU box: tests, the bench line, rocprofv3 kernel stats of the SAME bench command,
# the four PMC passes (tools/pmc_run.py) folded into one JSON
set -o pipefail
out=${1:-gpurun_out/round}
mkdir -p $out && export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; echo "gpu tests exit $?" >> $out/gpu_tests.log; tail -3 $out/gpu_tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python3 bench.py > $out/bench_under_rocprof.json 2> $out/rocprof.err; echo "rocprof exit $?"
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv; find $out -name "*kernel_trace.csv" -delete
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_$tag -o run --output-format csv -- python3 tools/pmc_run.py > $out/pmc_$tag.log 2>&1; echo "pmc $tag exit $?"
done
python tools/pmc_summarize.py $out/pmc_TCC_HIT_sum $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $out/pmc.json
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
head -9 $out/kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
