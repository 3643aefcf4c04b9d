#!/bin/bash
# train-step kernel profile (immediate mode), then the same step with MIOpen's find mode (cudnn.benchmark)
set -o pipefail
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( while true; do date > gpurun_out/heartbeat.txt; sleep 30; done ) & HB=$!
trap "kill $HB" EXIT
(cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/r3_train_prof -- python3 -m artemis_amd.train --model artemis --batch 8 --size 256 --steps 5 --device hip:0 --cudnn_benchmark 0 > $R/gpurun_out/r3k_train_prof.log 2>&1)
rc=$?; [ $rc -ne 0 ] && { echo "rocprof train rc=$rc"; tail -n 8 gpurun_out/r3k_train_prof.log; exit 1; }
grep "^step" gpurun_out/r3k_train_prof.log
python tools/train_step_profile.py /tmp/r3_train_prof gpurun_out/r3_train_step_kernel_stats.csv 15 | cut -c1-220
PYTHONPATH=$R timeout -k 10 100 python -m artemis_amd.train --model artemis --batch 8 --size 256 --steps 5 --device hip:0 --cudnn_benchmark 0 > gpurun_out/r3k_train_immediate.log 2>&1
grep "^step" gpurun_out/r3k_train_immediate.log
PYTHONPATH=$R timeout -k 10 600 python -m artemis_amd.train --model artemis --batch 8 --size 256 --steps 5 --device hip:0 --cudnn_benchmark 1 > gpurun_out/r3k_train_findmode.log 2>&1
echo "find mode rc=$?"; grep "^step" gpurun_out/r3k_train_findmode.log
