#!/usr/bin/env python3
"""Pipeline breakdown of the other configurations (default variant)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cs397raytracingsp22_amd import Context, scenes
cases = {"cfg1": lambda: scenes.config1(1920, 1080, 64, 8), "cfg4": lambda: scenes.config4(1920, 1080, 64, 10, tex_size=1024),
         "cfg5": lambda: scenes.config5(1920, 1080, 64, 50), "head": lambda: scenes.head_scene(800, 800, 64, 10, textures=scenes.load_asset_textures())}
for name in (sys.argv[1:] or list(cases)):
    sc = cases[name](); ctx = Context(0); ctx.upload(sc.flatten()); ctx.reserve(sc.camera)
    best = None
    for rep in range(3):
        _, _, _, st = ctx.render(sc.camera, want_u8=False)
        if best is None or st.kernel_ms < best[0]: best = (st.kernel_ms, ctx.last_pipeline_ms(), st.samples)
    print("RES %s: %.1f ms, %.0f Msamples/s" % (name, best[0], best[2] / best[0] / 1e3), {a: round(b, 1) for a, b in best[1].items()}, flush=True)
    ctx.close()
