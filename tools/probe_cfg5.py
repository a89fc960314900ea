#!/usr/bin/env python3
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes
ctx = Context(0)
for spp in (256, 1024, 4096):
    sc = scenes.config5(1920, 1080, spp, 50)
    ctx.upload(sc.flatten())
    ctx.reserve(sc.camera, 1)
    for rep in range(1 if spp > 1024 else 2):
        t0 = time.perf_counter()
        _, _, _, st = ctx.render(sc.camera, seed=1, want_u8=False, want_f32=False)
        wall = time.perf_counter() - t0
    print(spp, "kernel_ms %.0f wall_ms %.0f Msamples/s %.0f" % (st.kernel_ms, wall * 1e3, st.samples / st.kernel_ms / 1e3), ctx.last_pipeline_ms(), flush=True)
