"""`python -m cs397raytracingsp22_amd.run` — the reference's run() (tracing.rs:354-548) on the GPU:
build a scene from the reference's types, Scene::render_to_image(), save render.png.

    python -m cs397raytracingsp22_amd.run --scene head --width 800 --height 800 --spp 100 --out render.png
"""
from __future__ import annotations

import argparse
import time

from . import scenes
from .image import save_png


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="head", choices=["head", "cfg1", "cfg2", "cfg4", "cfg5"])
    ap.add_argument("--width", type=int, default=100)      # run()'s literal: 100 x 100, 100 spp, depth 10
    ap.add_argument("--height", type=int, default=100)
    ap.add_argument("--spp", type=int, default=100)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default="render.png")
    a = ap.parse_args(argv)
    if a.scene == "head":
        sc = scenes.head_scene(a.width, a.height, a.spp, a.depth, textures=scenes.load_asset_textures())
    elif a.scene == "cfg1":
        sc = scenes.config1(a.width, a.height, a.spp, a.depth)
    elif a.scene == "cfg2":
        sc = scenes.config2(a.width, a.height, a.spp, a.depth)
    elif a.scene == "cfg4":
        sc = scenes.config4(a.width, a.height, a.spp, a.depth)
    else:
        sc = scenes.config5(a.width, a.height, a.spp, a.depth)
    print("Rendering...")                                   # tracing.rs:222
    t0 = time.perf_counter()
    img = sc.render_to_image(seed=a.seed, device=a.device)
    print(f"Done. ({time.perf_counter() - t0:.2f} s)")      # tracing.rs:261
    save_png(a.out, img)                                    # tracing.rs:546
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
