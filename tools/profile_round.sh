# The round's evidence on the GPU box, one rocprofv3 pass per question (tools/profile_round.sh [outdir] [tag]):
#   1. (last) the bench line itself (python bench.py, no profiler), and the driver's short form --steps 20 --warmup 5
#   2. kernel statistics of the HEADLINE alone: rocprofv3 --kernel-trace --stats -- python3 bench.py --headline-only --steps 2048
#      --warmup 2048: every launch of the dominant kernel (N=256: wide_kernel<false, true, true>, 2048 iterations per launch) is
#      one the bench line's roofline block prices, so its average duration there is comparable with roofline.launch_us
#   3. kernel statistics of each secondary leg by itself (--leg farm | small_n | cu_batch)
#   4. four --pmc passes of tools/pmc_run.py (N=256, the solver's own path: one chip-wide launch of 64 iterations), folded into one JSON
set -o pipefail
out=${1:-gpurun_out/round}
tag=${2:-r04_vX}
mkdir -p $out && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_headline -o run -- python3 bench.py --headline-only --steps 2048 --warmup 2048 > $out/bench_headline_under_rocprof.json 2> $out/rocprof_headline.err; echo "rocprof headline exit $?"
find $out/prof_headline -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
for leg in farm small_n cu_batch; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$leg -o run -- python3 bench.py --leg $leg > $out/leg_${leg}_under_rocprof.json 2> $out/rocprof_$leg.err; echo "rocprof $leg exit $?"
  find $out/prof_$leg -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_$leg.csv
done
find $out -name "*kernel_trace.csv" -delete
for grp in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  ctr=$(echo $grp | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $grp -d $out/pmc_$ctr -o run --output-format csv -- python3 tools/pmc_run.py > $out/pmc_$ctr.log 2>&1; echo "pmc $ctr exit $?"
done
python tools/pmc_summarize.py $out/pmc_TCC_HIT_sum $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $out/pmc.json
find $out -name "*counter_collection.csv" -delete; find $out -name "*kernel_trace.csv" -delete
# the bench line LAST: its roofline.traffic comes from the newest profiles/r*_pmc.json -- the passes just made
cp $out/pmc.json profiles/${tag}_pmc.json
python bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"
python bench.py --steps 20 --warmup 5 --no-cpu > $out/bench_steps20.json 2> $out/bench_steps20.err; echo "bench --steps 20 exit $?"
head -6 $out/kernel_stats.csv | cut -d, -f1-4 | cut -c1-170
