#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
E=artemis_amd/lib/exp
timeout -k 10 300 python -u -m pytest tests/test_adaptive_radius.py tests/test_tile_configs.py "tests/test_baseline_configs.py" -x -q -m gpu --timeout 180 --timeout-method thread -k "adaptive or tile or config1 or regimes" > gpurun_out/r3l_tests_a.log 2>&1
rc=$?; tail -n 6 gpurun_out/r3l_tests_a.log | cut -c1-220; [ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/ab.py $E/libsynth_new.so --head --radius -3 > gpurun_out/r3l_ab_head_small.log 2>&1 || { tail -n 5 gpurun_out/r3l_ab_head_small.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3l_ab_head_small.log
timeout -k 10 200 python tools/ab.py $E/libsynth_new.so $E/libsynth_lean.so $E/libsynth_r2.so --head --radius 8 > gpurun_out/r3l_ab_head_r8.log 2>&1 || { tail -n 5 gpurun_out/r3l_ab_head_r8.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3l_ab_head_r8.log
timeout -k 10 200 python tools/ab.py $E/libsynth_new.so --radius -3 > gpurun_out/r3l_ab_op_small.log 2>&1 || { tail -n 5 gpurun_out/r3l_ab_op_small.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3l_ab_op_small.log
timeout -k 10 200 python tools/ab.py $E/libsynth_new.so $E/libsynth_lean.so $E/libsynth_r2.so --radius 8 > gpurun_out/r3l_ab_op_r8.log 2>&1 || { tail -n 5 gpurun_out/r3l_ab_op_r8.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3l_ab_op_r8.log
timeout -k 10 200 python tools/ab.py $E/libsynth_new.so --size 256,448 --batch 16 --radius -3 > gpurun_out/r3l_ab_config2_small.log 2>&1 || { tail -n 5 gpurun_out/r3l_ab_config2_small.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3l_ab_config2_small.log
EXP=exp_glds timeout -k 10 120 python tools/exp/run_exp.py 0,1,2,3,11,12,13 > gpurun_out/r3l_glds.log 2>&1 || { tail -n 5 gpurun_out/r3l_glds.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3l_glds.log
