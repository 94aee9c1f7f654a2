# quick check of a kernel change on the GPU box: a parity subset, rocprofv3 kernel stats of the bench, the bench line
out=${1:-gpurun_out/quick}
mkdir -p $out && export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_persistent.py -q -m gpu -x -k "trajectory or solve_converges or max_iterations or smoother or chunking" > $out/t.log 2>&1; tail -3 $out/t.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof -o run -- python3 bench.py --steps 1000 --warmup 100 --no-cpu --no-farm > $out/bench_rocprof.json 2> $out/rocprof.err
find $out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv; find $out -name "*kernel_trace.csv" -delete
head -8 $out/kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
python bench.py --no-cpu --no-farm | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d[k] for k in ('value','ms_per_step','step_only_value')}, d['roofline']['launch_us'])"
