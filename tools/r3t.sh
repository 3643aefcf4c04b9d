#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 280 python -u -m pytest tests/test_gin_gather.py -x -q -m gpu --timeout 120 --timeout-method thread > gpurun_out/r3t_gather_tests.log 2>&1
rc=$?; tail -n 3 gpurun_out/r3t_gather_tests.log | cut -c1-220; [ $rc -ne 0 ] && { tail -n 40 gpurun_out/r3t_gather_tests.log | cut -c1-200; exit 1; }
timeout -k 10 300 python tools/sweep.py --radii=-3 --gin --sigmas 0.045,1 > gpurun_out/r3t_sweep_gin.log 2>&1 || { echo sweep failed; tail -n 5 gpurun_out/r3t_sweep_gin.log; exit 1; }
grep -v amdgpu.ids gpurun_out/r3t_sweep_gin.log | cut -c1-150
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3t_prof -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py --gin 1 --radius -3 --sigma 0.045 --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/r3t_prof.log 2>&1) || { echo rocprof failed; tail -5 gpurun_out/r3t_prof.log; exit 1; }
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3t_prof1 -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py --gin 1 --radius -3 --sigma 1 --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/r3t_prof1.log 2>&1) || { echo rocprof failed; tail -5 gpurun_out/r3t_prof1.log; exit 1; }
python3 - <<'PY'
import csv, glob
for d in ('r3t_prof', 'r3t_prof1'):
    print(d)
    for f in sorted(glob.glob('gpurun_out/%s/**/*kernel_stats.csv' % d, recursive=True))[-1:]:
        for r in csv.DictReader(open(f)):
            n = r['Name']
            if 'synth' in n or 'finalize' in n or 'fillBuffer' in n or 'absmax' in n or 'zero_if' in n:
                print('  %-60s calls %4s avg %9.1f us min %9.1f' % (n.split('(anonymous namespace)::')[-1].split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
find gpurun_out/r3t_prof gpurun_out/r3t_prof1 -name "*kernel_trace.csv" -delete
