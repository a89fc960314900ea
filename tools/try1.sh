#!/bin/bash
python - <<PY
import sys; sys.path.insert(0,".")
import torch
from cs397raytracingsp22_amd import Context, scenes
sc = scenes.config2(1920,1080,256,10); ctx = Context(0); ctx.upload(sc.flatten()); ctx.reserve(sc.camera)
for i in range(4):
    _,_,_,st = ctx.render(sc.camera, want_u8=False, variant=7)
    print("RES %.1f ms" % st.kernel_ms, {k: round(v,1) for k,v in ctx.last_pipeline_ms().items()})
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
