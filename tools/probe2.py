#!/usr/bin/env python3
"""Developer probe: same 1080p/256spp frame with and without the teapot, both variants."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cs397raytracingsp22_amd import Context, scenes, abi
ctx = Context(0)
for name, sc in (("cornell-only 1080p/256", scenes.config1(1920, 1080, 256, 10)), ("cornell+teapot 1080p/256", scenes.config2(1920, 1080, 256, 10))):
    ctx.upload(sc.flatten())
    for vname, v in (("simple", abi.MI_VARIANT_SIMPLE), ("parked", abi.MI_VARIANT_PARKED)):
        best = 1e30
        for rep in range(2):
            _, _, _, st = ctx.render(sc.camera, seed=1, want_f32=True, want_u8=False, variant=v)
            best = min(best, st.kernel_ms)
        print(f"{name} {vname}: kernel_ms={best:.2f} Msamples/s={st.samples / best / 1e3:.1f}", flush=True)
