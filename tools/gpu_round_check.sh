set -o pipefail
mkdir -p gpurun_out/r02c
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r02c/gpu_tests.log 2>&1; echo "gpu tests exit $?" >> gpurun_out/r02c/gpu_tests.log
tail -6 gpurun_out/r02c/gpu_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r02c/bench.json 2> gpurun_out/r02c/bench.err; echo "bench exit $?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/r02c/bench_k20.json 2> gpurun_out/r02c/bench_k20.err; echo "bench k20 exit $?"
cat gpurun_out/r02c/bench.json gpurun_out/r02c/bench_k20.json
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02c/prof -o run -- python3 bench.py --steps 1000 --warmup 100 --no-cpu --no-farm > gpurun_out/r02c/bench_rocprof.json 2> gpurun_out/r02c/rocprof.err; echo "rocprof exit $?"
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $grp -d gpurun_out/r02c/pmc_$tag -o run --output-format csv -- python3 tools/pmc_run.py > gpurun_out/r02c/pmc_$tag.log 2>&1; echo "pmc $tag exit $?"
done
python tools/pmc_summarize.py gpurun_out/r02c/pmc_TCC_HIT_sum gpurun_out/r02c/pmc_FETCH_SIZE gpurun_out/r02c/pmc_WRITE_SIZE gpurun_out/r02c/pmc_SQ_VALU_MFMA_BUSY_CYCLES > gpurun_out/r02c/pmc.json
find gpurun_out/r02c/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r02c/kernel_stats.csv
# keep the merge small: drop the raw traces
find gpurun_out/r02c -name "*kernel_trace.csv" -delete; find gpurun_out/r02c -name "*counter_collection.csv" -size +20M -delete
head -12 gpurun_out/r02c/kernel_stats.csv
