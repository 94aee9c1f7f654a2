mkdir -p gpurun_out/r02k
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r02k/gpu_tests.log 2>&1; echo "gpu tests exit $?" >> gpurun_out/r02k/gpu_tests.log; tail -6 gpurun_out/r02k/gpu_tests.log
KB_MASKS=32768 timeout -k 10 300 python tools/kbench.py --reps 300 > gpurun_out/r02k/kbench_setprio.log 2>&1; grep -E "^stage|mask 32768" gpurun_out/r02k/kbench_setprio.log
