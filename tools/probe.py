#!/usr/bin/env python3
"""Quick GPU timing probe (developer tool): kernel ms per config and variant."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cs397raytracingsp22_amd import Context, scenes, abi

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = Context(0)
for name, sc in (("cfg1", scenes.config1(400, 400, 16, 8)), ("cfg2", scenes.config2(1920, 1080, spp, 10)),
                 ("cfg5", scenes.config5(1920, 1080, 64, 50))):
    ctx.upload(sc.flatten())
    for vname, v in (("simple", abi.MI_VARIANT_SIMPLE), ("parked", abi.MI_VARIANT_PARKED)):
        best = 1e30
        for rep in range(3):
            _, _, _, st = ctx.render(sc.camera, seed=1, want_f32=True, want_u8=False, variant=v)
            best = min(best, st.kernel_ms)
        print(f"{name} {vname}: samples={st.samples} kernel_ms={best:.2f} Msamples/s={st.samples / best / 1e3:.1f} lds={st.scene_in_lds}", flush=True)
