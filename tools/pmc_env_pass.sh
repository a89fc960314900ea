#!/bin/bash
# One rocprofv3 PMC pass of bench.py per developer-knob setting (MI_RT_* variables, exported here: the program itself follows `--`):
#   tools/pmc_env_pass.sh <out-subdir of gpurun_out> <config> "<counters>" "VAR=a" "VAR=b" ...   then  python tools/diag_sum.py gpurun_out/<subdir> [kernel]
TAG="$1"; CFG="$2"; CTRS="$3"; shift 3
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for setting in "$@"; do
  name="$(echo "$setting" | tr ' =' '__')"
  ( export $setting; timeout -k 10 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --config "$CFG" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1; echo "pass $name exit $?" )
done
