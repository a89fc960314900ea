#!/bin/bash
# one-off diagnostic PMC passes (program directly after --; --pmc with --kernel-trace only)
#   tools/diag_pmc.sh <out-subdir of gpurun_out> <bench config> [pass names...]
TAG="${1:-diag}"; CFG="${2:-cfg2}"; shift 2
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"; ROOT="$GRAFT_REPO_ROOT"
cd /tmp && export TMPDIR=/tmp
WANT=" $* "
pass() { local name="$1"; shift
  if [ "$WANT" != "  " ] && [[ "$WANT" != *" $name "* ]]; then return; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --config "$CFG" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1
  echo "pass $name exit $?"; }
pass sqa SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES
pass sqb SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VALU_TRANS_F32
pass sqc SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_CYCLES SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH SQ_INST_LEVEL_SMEM
pass ta TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
pass ta2 TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum
pass grbm GRBM_GUI_ACTIVE GRBM_TA_BUSY
