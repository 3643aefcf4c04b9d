#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r3r_rampup.log
: > $O
run() { echo "== $*" >> $O; timeout -k 10 170 "$@" 2>>gpurun_out/r3r.err | python3 -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('   value', d['value'], 'ms/step', d['ms_per_step'], 'bwd launch ms', r['avg_launch_ms'], 'frac', r['frac'], 'fwd ms', d['config'].get('fwd_kernel_ms'))
" >> $O || { echo "FAILED: $*"; tail -n 5 gpurun_out/r3r.err; exit 1; }; }
B="python bench.py --no-cpu-baseline --no-head --regimes="
run $B
run $B --steps 200 --warmup 100
run $B
run $B --steps 200 --warmup 100
run $B --steps 1000 --warmup 500
run $B --radius -3
run $B --radius -3 --steps 200 --warmup 100
cat $O
