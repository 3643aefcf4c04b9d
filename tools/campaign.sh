#!/bin/bash
# The measurement campaign behind profiles/r3_* (run on the GPU box from the repo root:
# gpurun -- 'bash tools/campaign.sh'); every GPU step under its own time limit, first failure ends it.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3z
mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { echo tests failed; tail -n 30 $O/tests.log; exit 1; }
tail -n 2 $O/tests.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
echo bench ok
timeout -k 10 200 python bench.py --workload interp7 > $O/bench_interp7.json 2> $O/bench_interp7.err || { echo interp7 failed; tail -5 $O/bench_interp7.err; exit 1; }
echo interp7 ok
timeout -k 10 200 python bench.py --workload config2 > $O/bench_config2.json 2> $O/bench_config2.err || { echo config2 failed; tail -5 $O/bench_config2.err; exit 1; }
echo config2 ok
timeout -k 10 500 python tools/sweep.py --radii=-3,-1,24 --gin --smooth > $O/sweep_gin.log 2>&1 || { echo sweep failed; tail -5 $O/sweep_gin.log; exit 1; }
echo sweep ok
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-head --regimes "" > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err) || { echo rocprof bench failed; tail -5 $O/prof_bench.err; exit 1; }
echo rocprof bench ok
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_head -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py --head --gin 1 --radius -3 > $GRAFT_REPO_ROOT/$O/prof_head.log 2>&1) || { echo rocprof head failed; tail -5 $O/prof_head.log; exit 1; }
echo rocprof head ok
tools/pmc_passes.sh $O/pmc --head --radius -3 || exit 1
PROG=tools/pmc_calib.py PASSES="fetch write" tools/pmc_passes.sh $O/pmc_calib || exit 1
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -size +4M -delete
echo all done
