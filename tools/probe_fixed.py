#!/usr/bin/env python3
"""Fixed per-launch cost of the pipeline kernels: a frame with almost no work."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes
for (w, h, spp) in ((64, 64, 4), (256, 256, 16), (512, 512, 64)):
    sc = scenes.config2(w, h, spp, 10)
    ctx = Context(0); ctx.upload(sc.flatten()); ctx.reserve(sc.camera)
    best = None
    for rep in range(5):
        _, _, _, st = ctx.render(sc.camera, want_u8=False)
        pm = ctx.last_pipeline_ms()
        if best is None or st.kernel_ms < best[0]: best = (st.kernel_ms, pm)
    print("RES %dx%dx%d: frame %.3f ms" % (w, h, spp, best[0]), {a: round(b, 3) for a, b in best[1].items()}, flush=True)
    ctx.close()
