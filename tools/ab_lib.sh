#!/bin/bash
# A/B of prebuilt library variants on ONE box (same process recipe, library swapped in between):
#   tools/ab_lib.sh "<probe.py args>" tmp_variants/libA.so tmp_variants/libB.so ...
# Build variants with e.g.  MI_RT_EXTRA_FLAGS=-DPT_MAIN_WAVES=6 bash cs397raytracingsp22_amd/csrc/build.sh && cp .../libmi_rt.so tmp_variants/libw6.so
ARGS="$1"; shift
cp cs397raytracingsp22_amd/lib/libmi_rt.so /tmp/libmi_rt_keep.so
for rep in 1 2; do for e in "$@"; do cp "$e" cs397raytracingsp22_amd/lib/libmi_rt.so; echo "RES [$e] $(python tools/probe.py $ARGS 2>&1 | grep -a '^RES' | tail -1)"; done; done
cp /tmp/libmi_rt_keep.so cs397raytracingsp22_amd/lib/libmi_rt.so
