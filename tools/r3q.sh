#!/bin/bash
# sustained (bench.py) vs launch-by-launch (tools/ab.py) timing of the N = 7 head forward and the N = 16 op forward
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r3q_sustained.log
: > $O
L=artemis_amd/lib/libartemis_synth.so
run() { echo "== $*" >> $O; timeout -k 10 170 "$@" 2>>gpurun_out/r3q.err | python3 -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('   value', d['value'], 'ms/step', d['ms_per_step'], 'launch ms', r['avg_launch_ms'], 'frac', r['frac'], 'adaptive', d['config'].get('adaptive'))
    elif 'fwd' in ln or 'bwd' in ln:
        print('   ' + ln)
" >> $O || { echo "FAILED: $*"; tail -n 5 gpurun_out/r3q.err; exit 1; }; }
B="python bench.py --no-cpu-baseline"
run $B --workload interp7
run $B --workload interp7 --steps 5 --warmup 2
run $B --workload interp7 --steps 60 --warmup 20
run $B --workload interp7 --radius -3
run $B --workload interp7 --radius -3 --steps 60 --warmup 20
run python tools/ab.py $L --head --radius -3 --batch 7 --rounds 6
run $B --workload config2
run $B --workload config2 --steps 200 --warmup 100
run $B --workload config2 --radius -3 --steps 200 --warmup 100
run python tools/ab.py $L --radius -3 --batch 16 --size 256,448 --rounds 10
cat $O
