#!/bin/bash
# End-of-round measurements on ONE box, plain (no profiler):  tools/final_probe.sh <outdir>
#   every BASELINE configuration at its stated size (tools/probe.py, best of 3), the 1/2, 1/4, 1/8 tile shares of cfg2 rendered rank by
#   rank (a rehearsal of the partition: no gather), and mi_multi_render with 2 / 8 ranks on this one device through the loopback test
#   transport (what the multi-rank machinery costs; not a scaling figure).
OUT="$1"; mkdir -p "$OUT"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd "$ROOT"
{
python tools/probe.py --config cfg1 --width 400 --height 400 --spp 16 --reps 5
python tools/probe.py --config cfg1
for cfg in cfg2 cfg3 cfg4 cfg5 head; do python tools/probe.py --config $cfg --reps 3; done
for w in 2 4 8; do python tools/probe.py --config cfg2 --world $w --reps 3; done
for n in 1 2 8; do python bench.py --gpus $n --loopback --steps 5 --warmup 1 --no-cpu-baseline | cut -c1-700; done
} 2>&1 | grep -a "^RES\|^{" > "$OUT/final_all_configs_probe.log"
cat "$OUT/final_all_configs_probe.log" | cut -c1-260
