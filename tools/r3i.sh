#!/bin/bash
# new kernels (segment pairing, reach statistics, adaptive radius): parity suite, bench lines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -u -m pytest tests/test_adaptive_radius.py tests/test_baseline_configs.py tests/test_tile_configs.py tests/test_geometry.py -x -q -m gpu --timeout 180 --timeout-method thread > gpurun_out/r3i_tests_a.log 2>&1
rc=$?; tail -n 25 gpurun_out/r3i_tests_a.log | cut -c1-220; [ $rc -ne 0 ] && exit 1
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err || { tail -n 20 gpurun_out/r3i_bench.err; exit 1; }
cat gpurun_out/r3i_bench.json
timeout -k 10 200 python bench.py --workload config2 --no-cpu-baseline > gpurun_out/r3i_bench_config2.json 2> gpurun_out/r3i_bench_config2.err || { tail -n 20 gpurun_out/r3i_bench_config2.err; exit 1; }
cat gpurun_out/r3i_bench_config2.json
timeout -k 10 200 python bench.py --workload interp7 > gpurun_out/r3i_bench_interp7.json 2> gpurun_out/r3i_bench_interp7.err || { tail -n 20 gpurun_out/r3i_bench_interp7.err; exit 1; }
cat gpurun_out/r3i_bench_interp7.json
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 420 --timeout-method thread --durations=10 --deselect tests/test_adaptive_radius.py --deselect tests/test_baseline_configs.py --deselect tests/test_tile_configs.py --deselect tests/test_geometry.py > gpurun_out/r3i_tests_b.log 2>&1
rc=$?; tail -n 30 gpurun_out/r3i_tests_b.log | cut -c1-200; [ $rc -ne 0 ] && exit 1
exit 0
