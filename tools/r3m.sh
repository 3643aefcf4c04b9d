#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
E=artemis_amd/lib/exp
timeout -k 10 200 python tools/ab.py $E/libsynth_lean.so $E/libsynth_w64a.so $E/libsynth_w64b.so --head --radius 8 > gpurun_out/r3m_ab_head_w64.log 2>&1 || { tail -n 5 gpurun_out/r3m_ab_head_w64.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3m_ab_head_w64.log
timeout -k 10 200 python tools/ab.py $E/libsynth_lean.so $E/libsynth_w64a.so $E/libsynth_w64b.so --radius 8 > gpurun_out/r3m_ab_op_w64.log 2>&1 || { tail -n 5 gpurun_out/r3m_ab_op_w64.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3m_ab_op_w64.log
EXP=exp_glds timeout -k 10 120 python tools/exp/run_exp.py 0,1,2,3,11,12,13 > gpurun_out/r3m_glds.log 2>&1 || { tail -n 12 gpurun_out/r3m_glds.log | cut -c1-200; exit 1; }
grep -v amdgpu.ids gpurun_out/r3m_glds.log
