#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
E=artemis_amd/lib/exp
timeout -k 10 300 python -u -m pytest tests/test_adaptive_radius.py tests/test_tile_configs.py tests/test_baseline_configs.py -x -q -m gpu --timeout 180 --timeout-method thread -k "adaptive or tile or config1 or regimes" > gpurun_out/r3n_tests_a.log 2>&1
rc=$?; tail -n 4 gpurun_out/r3n_tests_a.log | cut -c1-220; [ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" gpurun_out/r3n_tests_a.log | head -20; exit 1; }
timeout -k 10 200 python tools/ab.py $E/libsynth_dma.so $E/libsynth_new.so --head --radius -3 > gpurun_out/r3n_ab_head.log 2>&1 || { tail -n 5 gpurun_out/r3n_ab_head.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3n_ab_head.log
timeout -k 10 200 python tools/ab.py $E/libsynth_dma.so $E/libsynth_new.so --radius -3 > gpurun_out/r3n_ab_op.log 2>&1 || { tail -n 5 gpurun_out/r3n_ab_op.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3n_ab_op.log
timeout -k 10 200 python tools/ab.py $E/libsynth_dma.so $E/libsynth_new.so --size 256,448 --batch 16 --radius -3 > gpurun_out/r3n_ab_config2.log 2>&1 || { tail -n 5 gpurun_out/r3n_ab_config2.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3n_ab_config2.log
