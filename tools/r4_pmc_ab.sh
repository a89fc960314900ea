#!/bin/bash
# One PMC pass (sq2: wave-time split, LDS conflict counters) per library variant and configuration, on one box:
#   tools/r4_pmc_ab.sh <outdir> <config> <pass> libA.so libB.so ...
set -uo pipefail
OUT="$1"; CFG="$2"; PASS="$3"; shift 3
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cp "$ROOT/cs397raytracingsp22_amd/lib/libmi_rt.so" /tmp/libmi_rt_keep.so
for lib in "$@"; do
  name="$(basename "$lib" .so)"
  cp "$lib" "$ROOT/cs397raytracingsp22_amd/lib/libmi_rt.so"
  bash "$ROOT/tools/prof.sh" "$OUT/${CFG}_${name}" "$CFG" "$PASS" || { cp /tmp/libmi_rt_keep.so "$ROOT/cs397raytracingsp22_amd/lib/libmi_rt.so"; exit 1; }
done
cp /tmp/libmi_rt_keep.so "$ROOT/cs397raytracingsp22_amd/lib/libmi_rt.so"
