#!/usr/bin/env python3
"""What the top-level tree over the list's Triangles buys: Cornell walls + N small random Triangles + two spheres at 1080p / 16 spp,
default against MI_OPT_NO_LIST_TREE (every Triangle one by one); signatures must agree.   python tools/probe_list_tree.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import numpy as np
from cs397raytracingsp22_amd import Context, Lambertian, Metal, Scene, Triangle, abi, scenes

ctx = Context(0)
for n in [int(a) for a in sys.argv[1:]] or [40, 105, 400, 2000]:
    rng = np.random.default_rng(n)
    objs = scenes.cornell_walls() + scenes.cornell_spheres()
    for k in range(n):
        c = rng.uniform((-2.6, 0.2, -2.6), (2.6, 5.6, 2.6)); p = c + rng.uniform(-0.3, 0.3, (3, 3))
        mat = Lambertian(albedo=tuple(map(float, rng.uniform(0.2, 0.9, 3))), emission=(0, 0, 0)) if k % 3 else Metal(albedo=(0.8, 0.8, 0.8), emission=(0, 0, 0), roughness=0.2)
        objs.append(Triangle(tuple(map(float, p[0])), tuple(map(float, p[1])), tuple(map(float, p[2])), mat))
    sc = Scene(scenes.cornell_camera(1920, 1080, 16, 10), objs)
    ctx.upload(sc.flatten())
    res = {}
    for name, fl in (("tree", 0), ("one by one", abi.MI_OPT_NO_LIST_TREE)):
        best = None
        for _ in range(2):
            _, _, sig, st = ctx.render(sc.camera, seed=1, want_u8=False, want_f32=False, want_sig=False, flags=fl)
            best = st.kernel_ms if best is None else min(best, st.kernel_ms)
        _, _, sig, _ = ctx.render(sc.camera, seed=1, want_u8=False, want_f32=False, want_sig=True, flags=fl)
        res[name] = (best, sig)
    same = bool(np.array_equal(res["tree"][1], res["one by one"][1]))
    print(f"RES {n} small triangles + 10 walls + 2 spheres, 1080p x 16 spp: tree {res['tree'][0]:.2f} ms, one by one {res['one by one'][0]:.2f} ms "
          f"({res['one by one'][0] / res['tree'][0]:.2f} x), signatures equal: {same}", flush=True)
ctx.close()
