#!/bin/bash
# The whole profile of one configuration on the GPU box, in the order that keeps the committed files consistent:
#   tools/prof_all.sh <tag> <config> [extra bench args]
#   1. tools/prof.sh        rocprofv3 passes (kernel trace + stats; FETCH_SIZE, WRITE_SIZE, SQ counters each in a pass of its own)
#   2. tools/prof_json.py   profiles/<tag>_<config>_{pmc.json,kernel_stats.csv} and profiles/traffic_<config>.json
#   3. bench.py (plain)     profiles/<tag>_<config>_bench.log — run AFTER step 2, so its roofline block quotes the profile beside it
# Results land in gpurun_out/prof_<tag>_<config>/ and profiles/ of the box's copy; copy profiles/ back through gpurun_out/.
set -uo pipefail
TAG="$1"; CFG="$2"; shift 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof_${TAG}_${CFG}"
bash "$ROOT/tools/prof.sh" "$OUT" "$CFG" || exit 1
python3 "$ROOT/tools/prof_json.py" "$OUT" "$CFG" "$TAG" > "$OUT/prof_json.log" 2>&1 || { tail -5 "$OUT/prof_json.log"; exit 1; }
python3 "$ROOT/bench.py" --config "$CFG" --steps 10 --warmup 2 "$@" > "$ROOT/profiles/${TAG}_${CFG}_bench.log" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
mkdir -p "$ROOT/gpurun_out/profiles_${TAG}"
cp "$ROOT/profiles/${TAG}_${CFG}_"* "$ROOT/profiles/traffic_${CFG}.json" "$ROOT/gpurun_out/profiles_${TAG}/"
echo "profiled $CFG as $TAG"
