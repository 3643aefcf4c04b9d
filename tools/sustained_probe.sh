#!/bin/bash
# How long a timed region must be before the rate settles (profiles/r3_sustained.log): bench.py with
# short and long regions against launch-by-launch timing (tools/ab.py), for the N = 7 head forward
# (interp7), the N = 16 op forward (config2) and the headline step.  gpurun -- 'bash tools/sustained_probe.sh'
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/sustained.log
: > $O
L=artemis_amd/lib/libartemis_synth.so
run() { echo "== $*" >> $O; timeout -k 10 170 "$@" 2>>gpurun_out/sustained.err | python3 -c "
import sys, json
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith('{'):
        d = json.loads(ln); r = d['roofline']
        print('   value', d['value'], 'ms/step', d['ms_per_step'], 'launch ms', r['avg_launch_ms'], 'frac', r['frac'], 'fwd ms', d['config'].get('fwd_kernel_ms'), 'adaptive', d['config'].get('adaptive'))
    elif 'fwd' in ln or 'bwd' in ln:
        print('   ' + ln)
" >> $O || { echo "FAILED: $*"; tail -n 5 gpurun_out/sustained.err; exit 1; }; }
B="python bench.py --no-cpu-baseline"
for kw in "--steps 5 --warmup 2" "--steps 20 --warmup 5" "--steps 60 --warmup 20"; do
  run $B --workload interp7 $kw
  run $B --workload interp7 --radius -3 $kw
done
run python tools/ab.py $L --head --radius -3 --batch 7 --rounds 6
for kw in "--steps 20 --warmup 5" "--steps 200 --warmup 100"; do
  run $B --workload config2 $kw
  run $B --workload config2 --radius -3 $kw
done
run python tools/ab.py $L --radius -3 --batch 16 --size 256,448 --rounds 10
for kw in "--steps 20 --warmup 5" "--steps 200 --warmup 100" "--steps 1000 --warmup 500"; do
  run $B --no-head --regimes= $kw
  run $B --no-head --regimes= --radius -3 $kw
done
cat $O
