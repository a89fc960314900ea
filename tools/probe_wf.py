#!/usr/bin/env python3
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from cs397raytracingsp22_amd import Context, scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenes.config2(1920, 1080, spp, 10)
ctx = Context(0)
ctx.upload(sc.flatten())
for rep in range(2):
    _, _, _, st = ctx.render(sc.camera, seed=1, want_f32=True, want_u8=False, variant=7)
    print(f"wavefront: kernel_ms={st.kernel_ms:.2f} Msamples/s={st.samples / st.kernel_ms / 1e3:.1f}", flush=True)
