#!/bin/bash
# Build A/B variants of libartemis_synth.so into artemis_amd/lib/exp/ (developer tool).
# usage: tools/build_variants.sh name:"-DFOO=1 -DBAR=2" ...      (variants build one after another;
# each build compiles its five objects in parallel)
set -e
cd "$(dirname "$0")/.."
mkdir -p artemis_amd/lib/exp
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  python3 - "$name" $flags <<'PY'
import shutil, sys
sys.path.insert(0, ".")
from artemis_amd import build
name, flags = sys.argv[1], sys.argv[2:]
build.OBJ_DIR = build.OBJ_DIR + "_" + name
build.LIB_PATH = "artemis_amd/lib/exp/libsynth_%s.so" % name
print(build.build(force=True, extra_flags=flags))
PY
done
ls artemis_amd/lib/exp/
