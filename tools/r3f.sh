#!/bin/bash
# train-step tests (DDP over gloo with rehearsal_env, DDP over RCCL with one rank), the HIP wait repro,
# and the kernel-level profile of one batch-8 256^2 train step
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests/test_train_step.py tests/test_artemis_model.py -x -v -m gpu --timeout 420 --timeout-method thread > gpurun_out/r3f_train_tests.log 2>&1
rc=$?; tail -n 12 gpurun_out/r3f_train_tests.log; [ $rc -ne 0 ] && exit 1
for c in copy-idle kernel copy-same-thread copy-normal-priority copy; do
  timeout -k 10 60 python tools/dbg/hip_wait_repro.py --case $c >> gpurun_out/r3f_hip_wait_repro.log 2>&1
  rc=$?; [ $rc -eq 124 -o $rc -eq 137 -o $rc -eq 139 -o $rc -eq 134 ] && { echo "repro $c rc=$rc"; exit 1; }
done
AMD_DIRECT_DISPATCH=0 timeout -k 10 60 python tools/dbg/hip_wait_repro.py --case copy >> gpurun_out/r3f_hip_wait_repro.log 2>&1
grep "case" gpurun_out/r3f_hip_wait_repro.log
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_train_prof -- python3 -m artemis_amd.train --model artemis --batch 8 --size 256 --steps 5 --device hip:0 > $R/gpurun_out/r3f_train_prof.log 2>&1) || { echo rocprof train failed; tail -n 8 gpurun_out/r3f_train_prof.log; exit 1; }
python tools/train_step_profile.py gpurun_out/r3_train_prof gpurun_out/r3_train_step_kernel_stats.csv 15 | cut -c1-200
