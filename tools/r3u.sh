#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r3u_tests.log 2>&1
rc=$?; tail -n 4 gpurun_out/r3u_tests.log | cut -c1-220; [ $rc -ne 0 ] && { grep -n "FAILED\|Error\|assert" gpurun_out/r3u_tests.log | head -20; exit 1; }
timeout -k 10 300 python tools/ab.py artemis_amd/lib/libartemis_synth.so artemis_amd/lib/exp/libsynth_prev.so artemis_amd/lib/exp/libsynth_nocentre.so --head --radius -3 --rounds 10 > gpurun_out/r3u_head_ahead.log 2>&1 || { echo ab failed; tail -n 5 gpurun_out/r3u_head_ahead.log; exit 1; }
tail -n 7 gpurun_out/r3u_head_ahead.log
