#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the CPU oracle.

The reference holds no golden vectors and cannot be run, so these are
outputs of the ORACLE (oracle/, the plain-C restatement): windows of the configurations'
images as linear f32 radiance, u8 bytes and per-pixel path signatures.  They serve
  * CPU tests: the oracle binary keeps producing them bit for bit;
  * GPU tests: the HIP path reproduces them without the oracle in the loop.
Run:  python tools/make_golden.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cs397raytracingsp22_amd import scenes  # noqa: E402
from oracle import orc_py  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> (scene builder kwargs, window x0,y0,w,h, seed)
CASES = {
    "cfg1_400x400_16spp": (lambda: scenes.config1(400, 400, 16, 8), (168, 200, 64, 64), 1),
    "cfg2_1080p_256spp": (lambda: scenes.config2(1920, 1080, 256, 10), (912, 600, 64, 48), 1),
    "cfg2_defocus_480x270_16spp": (lambda: scenes.config2(480, 270, 16, 10, lens_radius=0.05), (200, 130, 96, 64), 7),
    "cfg4_drone_480x270_16spp": (lambda: scenes.config4(480, 270, 16, 10, tex_size=256), (180, 70, 128, 96), 1),
    "cfg5_subsurface_480x270_64spp_d50": (lambda: scenes.config5(480, 270, 64, 50), (190, 150, 96, 64), 1),
    "head_200x200_16spp": (lambda: scenes.head_scene(200, 200, 16, 10, textures=scenes.load_asset_textures()),
                           (20, 60, 160, 120), 1),
    # the rest of the Camera/Scene surface: ShadingMode::Phong (0) and CameraProjectionMode::Orthographic (0)
    "cfg2_phong_480x270_16spp": (lambda: scenes.with_camera(scenes.config2(480, 270, 16, 10), shading_mode=0,
                                                             point_light_pos=(0.5, 4.0, 2.5), ambient=(0.05, 0.1, 0.15)),
                                 (176, 110, 128, 96), 3),
    "cfg2_ortho_240x136_16spp": (lambda: scenes.with_camera(scenes.config2(240, 136, 16, 10), projection_mode=0),
                                 (56, 20, 128, 96), 1),
}


def main():
    for name, (mk, win, seed) in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        sc = mk()
        o = orc_py.OracleScene(sc.flatten())
        f32, u8, sig, _ = o.render(sc.camera, seed=seed, window=win)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), f32=f32, u8=u8, sig=sig,
                            window=np.int32(win), seed=np.int32(seed))
        print(name, f32.shape, "mean", f32.mean(axis=(0, 1)), "max", float(f32.max()))


if __name__ == "__main__":
    main()
