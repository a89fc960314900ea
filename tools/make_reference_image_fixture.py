#!/usr/bin/env python3
"""Reduce the reference's published render of its HEAD scene (render.png, 800x800 — an OUTPUT of the real Rust program,
the only one whose scene code is in HEAD, tracing.rs:356-543) to per-REGION statistics:

  tests/golden/reference_render_regions.npz
     labels    200x200 u8: region id of every 4x4-pixel block of the 800x800 image (0 = no region)
     names     region names, index = id - 1
     ref_mean  [R, 3] mean tone-mapped RGB (0..1) of render.png over the region
     ref_sigma [R]    per-pixel noise estimate of render.png inside the region (residual against a 3x3 box mean)
     n_pixels  [R]

Regions are cut from the scene literal, not from the picture: a pixel belongs to an object's region when the camera ray
through its centre hits that object first (the oracle's Scene::intersect_ray), eroded by 6 pixels so that anti-aliasing
footprints and silhouette differences stay out; the two ConvexVolumes (stochastic hits) use the projected disc of their
boundary sphere instead.  The drone is left out: its five TGA maps are missing from the reference (.MISSING_LARGE_BLOBS).
Also writes the older 50x50 block-mean fixture.  Data, not source.  Run in the build container:
    python tools/make_reference_image_fixture.py [/root/reference]"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cs397raytracingsp22_amd import scenes  # noqa: E402
from oracle import orc_py  # noqa: E402

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
img = np.asarray(Image.open(os.path.join(REF, "render.png")).convert("RGB")).astype(np.float32) / 255.0
assert img.shape == (800, 800, 3)
n = 50
small = img.reshape(n, 16, n, 16, 3).mean(axis=(1, 3)).astype(np.float32)
np.save(os.path.join(OUT, "reference_render_50x50.npy"), small)

# ---- first-hit object of every second pixel centre (400x400 grid of the 800x800 image) ----
sc = scenes.head_scene(800, 800, 1, 10, textures=scenes.load_asset_textures())
flat = sc.flatten()
o = orc_py.OracleScene(flat)
cam = sc.camera
W = H = 800
p = 1.0 / H
hit = np.full((400, 400), -1, np.int32)
eye = np.array(cam.eyepoint, np.float64)
for j in range(400):
    for i in range(400):
        x, y = 2 * i + 0.5, 2 * j + 0.5                      # between the 2x2 pixels of the block
        c = np.array([p * (x - 0.5 * W + 0.5), p * (0.5 + 0.5 * H - y), -cam.focal_length])
        d = c / np.linalg.norm(c)                            # view -z, up +y: camera space = world axes
        h = o.intersect(eye, d, t_min=0.001, t_max=cam.max_trace_dist)
        hit[j, i] = h.object if h.hit else -1
objs = sc.objects
names, masks = [], []


def erode(m, k):
    out = m.copy()
    for _ in range(k):
        s = out.copy()
        s[1:, :] &= out[:-1, :]; s[:-1, :] &= out[1:, :]; s[:, 1:] &= out[:, :-1]; s[:, :-1] &= out[:, 1:]
        s[0, :] = s[-1, :] = False; s[:, 0] = s[:, -1] = False
        out = s
    return out


kinds = [type(ob).__name__ for ob in objs]
pm = 0
for k, ob in enumerate(objs):
    kind = kinds[k]
    if kind == "StaticMesh":
        nm = ["drone", "cube_mesh", "sphere_mesh"][sum(1 for q in kinds[:k] if q == "StaticMesh")]
        if nm == "drone":
            continue
    elif kind == "Sphere":
        mat = type(ob.material).__name__
        if mat == "ParameterizedMaterial":
            nm = f"pm_sphere_rough{ob.material.roughness:g}_metal{ob.material.metallic:g}"; pm += 1
        elif mat == "Dielectric":
            nm = "glass_sphere_ior2.5"
        else:
            nm = "cyan_emitter"
    elif kind == "Plane":
        nm = "floor_plane"
    elif kind == "ConvexVolume":
        continue
    else:
        continue                                             # the light triangles are outside the view
    m = erode(hit == k, 3)                                   # 3 cells of the 400 grid = 6 pixels
    if nm == "floor_plane":
        m[:330, :] = False                                   # keep the near floor (below the horizon haze of distant reflections)
    if m.sum() >= 40:
        names.append(nm); masks.append(m)
# the two ConvexVolumes: projected disc of the boundary sphere, 55 % of its radius, clipped to the image
for k, ob in enumerate(objs):
    if kinds[k] != "ConvexVolume":
        continue
    c3 = np.array(ob.boundary.center, np.float64) - eye
    r = ob.boundary.radius
    cx = (c3[0] / -c3[2]) * cam.focal_length / p + 0.5 * W - 0.5
    cy = 0.5 + 0.5 * H - (c3[1] / -c3[2]) * cam.focal_length / p
    rad = 0.55 * r / -c3[2] * cam.focal_length / p
    yy, xx = np.mgrid[0:400, 0:400] * 2 + 0.5
    m = ((xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad) & (xx > 6) & (xx < W - 6)
    big = ((xx - cx) ** 2 + (yy - cy) ** 2 <= (2.0 * rad + 8) ** 2)          # the whole boundary disc + margin: nobody else's region
    for q in range(len(masks)):
        masks[q] = masks[q] & ~big
    names.append(f"volume_density{ob.density:g}"); masks.append(m)

labels400 = np.zeros((400, 400), np.uint8)
for r, m in enumerate(masks):
    assert not (labels400[m] != 0).any(), names[r]
    labels400[m] = r + 1
# 4x4-pixel blocks: a block is labelled when its 2x2 cells of the 400 grid agree
l4 = labels400.reshape(200, 2, 200, 2)
same = (l4 == l4[:, :1, :, :1]).all(axis=(1, 3))
labels = np.where(same, l4[:, 0, :, 0], 0).astype(np.uint8)
full = np.kron(labels, np.ones((4, 4), np.uint8))
box = img.copy()
acc = np.zeros_like(img); cnt = np.zeros((800, 800, 1), np.float32)
for dy in (-1, 0, 1):
    for dx in (-1, 0, 1):
        acc[max(0, dy):800 + min(0, dy), max(0, dx):800 + min(0, dx)] += img[max(0, -dy):800 + min(0, -dy), max(0, -dx):800 + min(0, -dx)]
        cnt[max(0, dy):800 + min(0, dy), max(0, dx):800 + min(0, dx)] += 1
resid = img - acc / cnt
ref_mean, ref_sigma, npx = [], [], []
for r in range(len(names)):
    m = full == r + 1
    ref_mean.append(img[m].mean(axis=0)); ref_sigma.append(float(resid[m].std() * np.sqrt(9.0 / 8.0))); npx.append(int(m.sum()))
    print(f"{r + 1:2d} {names[r]:38s} n={npx[-1]:6d} mean={np.round(ref_mean[-1], 3)} sigma={ref_sigma[-1]:.3f}")
np.savez_compressed(os.path.join(OUT, "reference_render_regions.npz"), labels=labels, names=np.array(names),
                    ref_mean=np.array(ref_mean, np.float32), ref_sigma=np.array(ref_sigma, np.float32), n_pixels=np.array(npx, np.int32))
