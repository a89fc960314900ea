#!/bin/bash
# rocprofv3 passes of bench.py on the GPU box (one parametrised script; replaces round 1's pmc*.sh):
#   tools/prof.sh <outdir> <config> [pass ...]        passes: trace fetch write sq1 sq2 grbm (default: all)
# Each counter set runs in its OWN pass with --kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots": FETCH_SIZE and
# WRITE_SIZE do not fit one pass; --pmc is never combined with other trace domains).  The program follows `--` directly.
# Then: python tools/prof_json.py <outdir> <config> <tag>   (writes profiles/<tag>_<config>_*.{json,csv})
set -uo pipefail
OUT="$1"; CFG="$2"; shift 2
PASSES=("$@"); [ ${#PASSES[@]} -eq 0 ] && PASSES=(trace fetch write sq1 sq2 grbm)
mkdir -p "$OUT"; OUT="$(cd "$OUT" && pwd)"
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
cd /tmp && export TMPDIR=/tmp
pmc() { local name="$1"; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- \
     python3 "$ROOT/bench.py" --config "$CFG" --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/$name.log" 2>&1
  local rc=$?; echo "pass $name exit $rc"; return $rc; }
for p in "${PASSES[@]}"; do
  case "$p" in
    trace) timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
             python3 "$ROOT/bench.py" --config "$CFG" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/trace.log" 2>&1; echo "pass trace exit $?" ;;
    fetch) pmc fetch FETCH_SIZE || exit 1 ;;
    write) pmc write WRITE_SIZE || exit 1 ;;
    sq1)   pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS || exit 1 ;;
    sq2)   pmc sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1 ;;
    grbm)  pmc grbm GRBM_GUI_ACTIVE || exit 1 ;;
  esac
done
