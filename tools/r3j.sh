#!/bin/bash
# A/B of the head forward variants + config2 shape; train-step profile (with a heartbeat: MIOpen's find
# mode spends minutes in the first step without printing)
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
E=artemis_amd/lib/exp
timeout -k 10 200 python tools/ab.py $E/libsynth_cur.so $E/libsynth_nodedupe.so $E/libsynth_passes1.so $E/libsynth_r2.so --head --radius 8 > gpurun_out/r3j_ab_head_r8.log 2>&1 || { tail -n 5 gpurun_out/r3j_ab_head_r8.log; exit 1; }
cat gpurun_out/r3j_ab_head_r8.log | grep -v amdgpu.ids
timeout -k 10 200 python tools/ab.py $E/libsynth_cur.so $E/libsynth_nodedupe.so $E/libsynth_passes1.so $E/libsynth_r2.so --head --radius 14 > gpurun_out/r3j_ab_head_r14.log 2>&1 || { tail -n 5 gpurun_out/r3j_ab_head_r14.log; exit 1; }
cat gpurun_out/r3j_ab_head_r14.log | grep -v amdgpu.ids
timeout -k 10 200 python tools/ab.py $E/libsynth_cur.so $E/libsynth_r2.so --size 256,448 --batch 16 --radius 14 > gpurun_out/r3j_ab_config2.log 2>&1 || { tail -n 5 gpurun_out/r3j_ab_config2.log; exit 1; }
cat gpurun_out/r3j_ab_config2.log | grep -v amdgpu.ids
timeout -k 10 200 python tools/ab.py $E/libsynth_cur.so $E/libsynth_passes1.so $E/libsynth_r2.so --radius 8 > gpurun_out/r3j_ab_op_r8.log 2>&1 || { tail -n 5 gpurun_out/r3j_ab_op_r8.log; exit 1; }
cat gpurun_out/r3j_ab_op_r8.log | grep -v amdgpu.ids
( while true; do date > gpurun_out/heartbeat.txt; sleep 30; done ) & HB=$!
(cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_train_prof -- python3 -m artemis_amd.train --model artemis --batch 8 --size 256 --steps 5 --device hip:0 > $R/gpurun_out/r3j_train_prof.log 2>&1)
rc=$?; kill $HB
[ $rc -ne 0 ] && { echo "rocprof train rc=$rc"; tail -n 8 gpurun_out/r3j_train_prof.log; exit 1; }
grep "^step" gpurun_out/r3j_train_prof.log
python tools/train_step_profile.py gpurun_out/r3_train_prof gpurun_out/r3_train_step_kernel_stats.csv 15 | cut -c1-200
rm -rf gpurun_out/r3_train_prof
