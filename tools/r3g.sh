#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
L=gpurun_out/r3h_hip_wait_repro.log
rep() { timeout -k 10 60 python tools/dbg/hip_wait_repro.py "$@" >> $L 2>&1; rc=$?; [ $rc -eq 124 -o $rc -eq 137 -o $rc -eq 139 -o $rc -eq 134 ] && { echo "repro $* rc=$rc"; grep case $L; exit 1; }; return 0; }
rep --case copy --prior 3
rep --case copy --prior 4
rep --case kernel --prior 4
rep --case copy-normal-priority --prior 4
AMD_DIRECT_DISPATCH=0 rep --case copy --prior 4
GPU_MAX_HW_QUEUES=8 rep --case copy --prior 4
GPU_MAX_HW_QUEUES=8 rep --case copy --prior 8
grep "case" $L
(cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_train_prof -- python3 -m artemis_amd.train --model artemis --batch 8 --size 256 --steps 5 --device hip:0 > $R/gpurun_out/r3g_train_prof.log 2>&1) || { echo rocprof train failed; tail -n 8 gpurun_out/r3g_train_prof.log; exit 1; }
grep "^step" gpurun_out/r3g_train_prof.log
python tools/train_step_profile.py gpurun_out/r3_train_prof gpurun_out/r3_train_step_kernel_stats.csv 15 | cut -c1-200
timeout -k 10 300 python -u -m pytest tests/test_train_step.py tests/test_artemis_model.py -x -q -m gpu --timeout 420 --timeout-method thread --durations=8 > gpurun_out/r3g_train_tests.log 2>&1
rc=$?; tail -n 14 gpurun_out/r3g_train_tests.log; [ $rc -ne 0 ] && exit 1
exit 0
