#!/bin/bash
# Throughput against the path-state budget (mi_render_opts.max_state_bytes) on the current pipeline:
#   tools/budget_curve.sh <outdir> [config]
# One bench.py run per budget, in one box, plain (no profiler); the JSON lines land in <outdir>/budget_<gb>.json.
set -uo pipefail
OUT="$1"; CFG="${2:-cfg2}"
mkdir -p "$OUT"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
for gb in 0 64 32 16 8 4 0; do
  timeout -k 10 300 python3 "$ROOT/bench.py" --config "$CFG" --steps 5 --warmup 1 --no-cpu-baseline --max-state-gb "$gb" > "$OUT/budget_${CFG}_${gb}_$RANDOM.json" 2> "$OUT/budget_${CFG}_${gb}.err" || { echo "budget $gb failed"; tail -3 "$OUT/budget_${CFG}_${gb}.err"; exit 1; }
  echo "budget $gb GB done"
done
python3 - "$OUT" "$CFG" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(f"{sys.argv[1]}/budget_{sys.argv[2]}_*.json")):
    for line in open(f):
        if line.startswith("{"):
            r = json.loads(line)
            print(f, r["config"]["max_state_gb"], round(r["ms_per_step"], 2), "ms", round(r["value"], 1), "Msamples/s", r["roofline"]["per_step_ms"])
PY
