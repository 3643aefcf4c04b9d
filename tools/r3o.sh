#!/bin/bash
# full GPU suite on the current library, then bench lines and the regime sweep
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 420 --timeout-method thread --durations=8 > gpurun_out/r3o_tests.log 2>&1
rc=$?; tail -n 16 gpurun_out/r3o_tests.log | cut -c1-200; [ $rc -ne 0 ] && { grep -n "Error\|FAILED" gpurun_out/r3o_tests.log | head; exit 1; }
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3o_bench.json 2> gpurun_out/r3o_bench.err || { tail -n 20 gpurun_out/r3o_bench.err; exit 1; }
cut -c1-1500 gpurun_out/r3o_bench.json
timeout -k 10 200 python bench.py --workload interp7 > gpurun_out/r3o_bench_interp7.json 2> gpurun_out/r3o_bench_interp7.err || { tail -n 20 gpurun_out/r3o_bench_interp7.err; exit 1; }
cut -c1-400 gpurun_out/r3o_bench_interp7.json
timeout -k 10 200 python bench.py --workload config2 --no-cpu-baseline > gpurun_out/r3o_bench_config2.json 2> gpurun_out/r3o_bench_config2.err || { tail -n 20 gpurun_out/r3o_bench_config2.err; exit 1; }
cut -c1-300 gpurun_out/r3o_bench_config2.json
timeout -k 10 500 python tools/sweep.py --radii=-3,-1 --sigmas 0.045,1,4 --gin > gpurun_out/r3o_sweep_gin.log 2>&1 || { tail -n 5 gpurun_out/r3o_sweep_gin.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3o_sweep_gin.log
