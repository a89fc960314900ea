#!/bin/bash
# What do SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU / SQ_BUSY_CYCLES report for kernels of ONE instruction class each (tools/microbench/valu_rates)?
OUT="${GRAFT_REPO_ROOT:-.}/gpurun_out/$1"; mkdir -p "$OUT"; BIN="${GRAFT_REPO_ROOT:-.}/tools/microbench/_build/valu_rates"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/calib" -- "$BIN" > "$OUT/calib.log" 2>&1
echo "exit $?"
