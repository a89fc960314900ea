#!/bin/bash
# per-launch durations of one frame of the default bench: tools/trace_frame.sh <outdir>
set -uo pipefail
OUT="$1"; mkdir -p "$OUT"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/trace.log" 2>&1; echo "trace exit $?"
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'wf_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows[-22:]:
    print("LAUNCH", r['Kernel_Name'][:20].ljust(22), round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 2), r.get('Grid_Size_X', ''))
PY
