#!/bin/bash
# gather-form checks after a change to the kernel: its tests, then gradInput's time at sigma 0.045
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 280 python -u -m pytest tests/test_gin_gather.py tests/test_zz_no_alloc.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/gin_chk_tests.log 2>&1
rc=$?; tail -n 1 gpurun_out/gin_chk_tests.log | cut -c1-200; [ $rc -ne 0 ] && { tail -n 30 gpurun_out/gin_chk_tests.log | cut -c1-200; exit 1; }
timeout -k 10 200 python tools/sweep.py --radii=-3 --gin --sigmas 0.045 2>&1 | grep "all\|gin " | cut -c1-130
