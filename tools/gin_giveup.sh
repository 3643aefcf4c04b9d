#!/bin/bash
# kernel time of the gather kernel when it gives up (sigma = 1), from a rocprofv3 kernel trace
set -o pipefail
O=gpurun_out/giveup; rm -rf $O; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -- python3 $GRAFT_REPO_ROOT/tools/prof_run.py --head --gin 1 --radius -3 > $GRAFT_REPO_ROOT/$O/run.log 2>&1) || { tail -5 $O/run.log; exit 1; }
find $O -name "*kernel_stats.csv" | head -1 | xargs grep -i "gin_gather" | awk -F'","|",' '{print "gather give-up: calls", $2, "avg ns", $4, "min", $6}'
