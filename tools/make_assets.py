#!/usr/bin/env python3
"""Convert the reference's OBJ input assets (obj/*.obj — input DATA)
into compact indexed-mesh .npz files with this repo's own OBJ reader
(cs397raytracingsp22_amd/objload.py, tobj single_index + triangulate semantics).

The reference tree does not exist on the GPU box, so the meshes the benchmark scenes
need travel as data under cs397raytracingsp22_amd/assets/.  Run here, in the build
container:   python tools/make_assets.py [/root/reference]
"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cs397raytracingsp22_amd import objload  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cs397raytracingsp22_amd", "assets")

for name in ("teapot", "cube", "drone", "sphere"):
    models = objload.load_obj(os.path.join(REF, "obj", name + ".obj"))
    m = models[0]           # the reference keeps models.remove(0) only (geometry.rs:157)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), positions=m.positions, normals=m.normals,
                        texcoords=m.texcoords, indices=m.indices)
    print(f"{name}: models={len(models)} vertices={m.n_vertices} triangles={m.n_triangles} "
          f"normals={m.normals.size // 3} texcoords={m.texcoords.size // 2}")

# textures the reference's own run() scene binds (tracing.rs:387-401): decoded here with PIL
# (image-file decoding is load-time work outside the path) and stored as RGB8 arrays.
try:
    from PIL import Image
    tex = {}
    for key, fn in (("green", "green.png"), ("magenta", "magenta.jpg"), ("normal_test_jpg", "normal_test.jpg"),
                    ("normal_test_png", "normal_test.png")):
        with Image.open(os.path.join(REF, "texture", fn)) as im:
            tex[key] = np.asarray(im.convert("RGB"))
        print(f"texture {key}: {tex[key].shape}")
    np.savez_compressed(os.path.join(OUT, "textures.npz"), **tex)
except ImportError:
    print("PIL missing: textures.npz not regenerated")
