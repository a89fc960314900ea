#!/bin/bash
# lane-utilisation + wave-time PMC passes for the default pipeline: tools/pmc3.sh <outdir>
set -uo pipefail
OUT="$1"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
run() { local name="$1"; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
     python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "$name exit $?"; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
