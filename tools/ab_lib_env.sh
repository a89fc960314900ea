#!/bin/bash
# A/B over library variants x developer-knob settings on ONE box:
#   tools/ab_lib_env.sh "<probe.py args>" "libA.so libB.so" "VAR=1" "VAR=2" ...
ARGS="$1"; LIBS="$2"; shift 2
cp cs397raytracingsp22_amd/lib/libmi_rt.so /tmp/libmi_rt_keep.so
for rep in 1 2; do for l in $LIBS; do cp "$l" cs397raytracingsp22_amd/lib/libmi_rt.so; for e in "$@"; do echo "RES [$l $e] $(env $e python tools/probe.py $ARGS 2>&1 | grep -a '^RES' | head -1)"; done; done; done
cp /tmp/libmi_rt_keep.so cs397raytracingsp22_amd/lib/libmi_rt.so
