#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=artemis_amd/lib/libartemis_synth.so
for b in 1 2 4 7; do
  timeout -k 10 200 python tools/ab.py $L --head --radius -3 --batch $b --rounds 6 2>&1 | grep "head fwd\|head bwd" | sed "s/^/B=$b /"
done
