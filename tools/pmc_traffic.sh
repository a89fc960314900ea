#!/bin/bash
# HBM traffic + kernel-trace passes of the default bench command (run on the GPU box):
#   tools/pmc_traffic.sh <outdir>
# Separate --pmc passes for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), each with
# --kernel-trace only, as MI355X_MICROARCH.md "rocprofv3 PMC slots" prescribes.
set -uo pipefail
OUT="$1"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/trace.log" 2>&1; echo "trace exit $?"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/fetch.log" 2>&1; echo "fetch exit $?"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/write.log" 2>&1; echo "write exit $?"
