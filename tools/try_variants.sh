#!/bin/bash
# developer: time pre-built library variants from tmp_variants/ (run on the GPU box)
for f in tmp_variants/libmi_rt_*.so; do
  cp "$f" cs397raytracingsp22_amd/lib/libmi_rt.so
  for r in ${REFILLS:-16}; do
  echo "$(basename $f) refill=$r: $(MI_RT_WF_REFILL=$r python - <<PY
import sys; sys.path.insert(0,".")
import torch
from cs397raytracingsp22_amd import Context, scenes
sc = scenes.config2(1920,1080,256,10); ctx = Context(0); ctx.upload(sc.flatten())
for i in range(3):
    _,_,_,st = ctx.render(sc.camera, want_u8=False, variant=7)
print("%.1f ms" % st.kernel_ms, {k: round(v,1) for k,v in ctx.last_pipeline_ms().items()})
PY
)"
  done
done
