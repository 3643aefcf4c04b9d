set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { echo bench failed; tail -5 $O/bench.err; exit 1; }
echo bench ok
timeout -k 10 200 python bench.py --workload interp7 > $O/bench_interp7.json 2> $O/bench_interp7.err || { echo interp7 failed; tail -5 $O/bench_interp7.err; exit 1; }
echo interp7 ok
timeout -k 10 500 python tools/sweep.py --radii=-1,0,4,8,24 --gin --smooth > $O/sweep.log 2>&1 || { echo sweep failed; tail -5 $O/sweep.log; exit 1; }
echo sweep ok
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_bench -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-head --regimes "" > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/prof_bench.err) || { echo rocprof bench failed; tail -5 $O/prof_bench.err; exit 1; }
echo rocprof bench ok
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_head -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py --head --gin 1 > $GRAFT_REPO_ROOT/$O/prof_head.log 2>&1) || { echo rocprof head failed; tail -5 $O/prof_head.log; exit 1; }
echo rocprof head ok
tools/pmc_passes.sh $O/pmc --head || exit 1
PROG=tools/pmc_calib.py PASSES="fetch write" tools/pmc_passes.sh $O/pmc_calib || exit 1
python tools/pmc_summary.py $O/pmc > $O/pmc_summary.txt
echo all done
