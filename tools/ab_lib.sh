#!/bin/bash
# A/B of prebuilt library variants on ONE box: tools/ab_lib.sh tmp_variants/libA.so tmp_variants/libB.so ...
run() { cp "$1" cs397raytracingsp22_amd/lib/libmi_rt.so; echo "RES [$1] $(python - <<PY
import sys; sys.path.insert(0,".")
import torch
from cs397raytracingsp22_amd import Context, scenes
sc = scenes.config2(1920,1080,256,10); ctx = Context(0); ctx.upload(sc.flatten()); ctx.reserve(sc.camera)
best = None
for i in range(4):
    _,_,_,st = ctx.render(sc.camera, want_u8=False, variant=7)
    p = ctx.last_pipeline_ms()
    if best is None or st.kernel_ms < best[0]: best = (st.kernel_ms, p)
print("%.1f ms" % best[0], {k: round(v,1) for k,v in best[1].items()})
PY
)"; }
for rep in 1 2 3 4; do for e in "$@"; do run "$e"; done; done
