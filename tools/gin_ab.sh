#!/bin/bash
set -o pipefail
for lib in "$@"; do
  echo "== $lib"
  ARTEMIS_SYNTH_LIB=$lib timeout -k 10 200 python tools/sweep.py --radii=-3 --gin --sigmas 0.045 2>&1 | grep "gin atomic" | cut -c1-130 || exit 1
done
