#!/bin/bash
# Build A/B variants of libartemis_synth.so into artemis_amd/lib/exp/ (developer tool).
# usage: tools/build_variants.sh name:"-DFOO=1 -DBAR=2" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p artemis_amd/lib/exp
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -munsafe-fp-atomics -ffp-contract=on -Iinclude $flags \
     -o artemis_amd/lib/exp/libsynth_$name.so artemis_amd/csrc/synth_kernels.hip 2>/dev/null &
done
wait
ls artemis_amd/lib/exp/
