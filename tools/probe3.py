#!/usr/bin/env python3
"""Developer probe: headline frame, all kernel variants + diagnostic counters of VOTED."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cs397raytracingsp22_amd import Context, scenes, abi
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
sc = {"cfg2": lambda: scenes.config2(1920, 1080, spp, 10), "cfg1": lambda: scenes.config1(1920, 1080, spp, 10),
      "cfg4": lambda: scenes.config4(1920, 1080, spp, 10), "cfg5": lambda: scenes.config5(1920, 1080, spp, 50),
      "head": lambda: scenes.head_scene(1920, 1080, spp, 10, textures=scenes.load_asset_textures())}[which]()
ctx = Context(0)
ctx.upload(sc.flatten())
for vname, v in (("voted", 3), ("wavefront", 7)):
    best = 1e30
    for rep in range(2):
        _, _, _, st = ctx.render(sc.camera, seed=1, want_f32=True, want_u8=False, variant=v)
        best = min(best, st.kernel_ms)
    print(f"{which} {vname}: kernel_ms={best:.2f} Msamples/s={st.samples / best / 1e3:.1f}", flush=True)
for dv in (4, 6):
    ctx.render(sc.camera, seed=1, want_f32=True, want_u8=False, variant=dv)
    d = ctx.last_diag()
    print("diag variant", dv, d)
    for k in ("a", "inner", "leaf"):
        t, l = d[k + "_trips"], d[k + "_lanes"]
        print(f"  {k}: trips/wave={t / max(1, d['waves']):.0f} active-lane fraction={l / max(1, 64 * t):.3f}")
    if d.get("shade_cycles"):
        tot = d["shade_cycles"] + d["gen_cycles"] + d["list_cycles"]
        print(f"  A split: shade {d['shade_cycles'] / tot:.2f}  gen {d['gen_cycles'] / tot:.2f}  list+root {d['list_cycles'] / tot:.2f}  (per trip {d['shade_cycles'] / d['a_trips']:.0f} / {d['gen_cycles'] / d['a_trips']:.0f} / {d['list_cycles'] / d['a_trips']:.0f} cycles)")
    if d.get("a_cycles"):
        print(f"  cycles: A {d['a_cycles'] / d['waves'] / 1e6:.2f} M/wave ({d['a_cycles'] / max(1, d['a_trips']):.0f} per trip)  "
              f"B {d['b_cycles'] / d['waves'] / 1e6:.2f} M/wave ({d['b_cycles'] / max(1, d['inner_trips'] + d['leaf_trips']):.0f} per micro-step)")
