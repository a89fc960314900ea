#!/bin/bash
# Build libmi_rt.so (HIP kernels for gfx950 + the C-ABI host runtime) in-tree.
# hipcc cross-compiles gfx950 without a GPU.  -ffp-contract=off: the reference (Rust)
# never fuses a*b+c, and bit-identical path decisions are what the parity tests check.
# -fno-slp-vectorize: the SLP vectorizer turns pairs of f32 ops into v_pk_* plus register moves;
# measured on wf_main: same VALU count, 240 more v_mov, 89 -> 81 ms without it.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/../lib"
mkdir -p "$OUT"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function ${MI_RT_EXTRA_FLAGS:-}"
"$HIPCC" $FLAGS -c "$HERE/pt_kernels.hip" -o "$OUT/pt_kernels.o"
"$HIPCC" $FLAGS -x hip --cuda-host-only -c "$HERE/mi_rt.cpp" -o "$OUT/mi_rt.o"
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$OUT/libmi_rt.so" "$OUT/pt_kernels.o" "$OUT/mi_rt.o" -ldl -lpthread
rm -f "$OUT/pt_kernels.o" "$OUT/mi_rt.o"
# C++ caller of the C ABI through the host mirror of the reference interface (host/*.hpp)
g++ -std=c++17 -O2 -ffp-contract=off -Wall -I"$HERE/../../include" "$HERE/../host/mi_rt_cli.cpp" -o "$OUT/mi_rt_cli" -L"$OUT" -lmi_rt -lz -Wl,-rpath,'$ORIGIN'
echo "built $OUT/libmi_rt.so and $OUT/mi_rt_cli"
