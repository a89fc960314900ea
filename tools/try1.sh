#!/bin/bash
MI_RT_DEBUG_MASK=1 python - <<PY
import sys; sys.path.insert(0,".")
import torch
from cs397raytracingsp22_amd import Context, scenes
sc = scenes.config2(1920,1080,256,10); ctx = Context(0); ctx.upload(sc.flatten()); ctx.reserve(sc.camera)
for i in range(2):
    _,_,_,st = ctx.render(sc.camera, want_u8=False, variant=7)
    print("RES %.1f ms" % st.kernel_ms, {k: round(v,1) for k,v in ctx.last_pipeline_ms().items()})
PY
