#!/bin/bash
# rocprofv3 PMC passes for the synth kernels (run on the GPU box from the repo root).
# One counter group per run (MI355X_MICROARCH.md: 8 SQ / 4 TCC slots; FETCH_SIZE=3, WRITE_SIZE=2).
# usage: [PROG=tools/pmc_calib.py] [PASSES="fetch write"] tools/pmc_passes.sh <outdir> [program args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
PROG=${PROG:-tools/prof_run.py}
PASSES=${PASSES:-"fetch write tcc sq1 sq2"}
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  case " $PASSES " in *" $name "*) ;; *) return 0;; esac
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/$PROG" $EXTRA > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  echo "pass $name ok"
}
EXTRA="$*"
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES
# (not in the default list: address translation, PASSES="tlb tlbstall")
run tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum
run tlbstall TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
echo done
