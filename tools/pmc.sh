#!/bin/bash
# rocprofv3 PMC passes for K1 (run on the GPU box through gpurun): each counter set in its
# own run with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots").
# usage: tools/pmc.sh <outdir> [bench args...]
set -uo pipefail
OUT="$1"; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name="$1"; shift
  timeout -k 10 280 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
     python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline "${BENCH_ARGS[@]}" > "$OUT/$name.log" 2>&1
  echo "$name exit $?"
}
BENCH_ARGS=("$@")
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
