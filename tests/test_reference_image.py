"""The one pin against an OUTPUT OF THE REAL REFERENCE PROGRAM, per region.

The reference publishes render.png (800x800) of the scene its run() builds (tracing.rs:356-543).  The scene is rebuilt
from that literal (scenes.head_scene) and compared REGION BY REGION — each of the 15 ParameterizedMaterial spheres
(:406-481), the ior-2.5 glass sphere (:487), the cyan emitter, both ConvexVolumes (:497-516), the floor Plane (:519), the
cube and sphere meshes — against region means of render.png (tests/golden/reference_render_regions.npz, made by
tools/make_reference_image_fixture.py; regions are cut from the scene literal by first-hit object, eroded 6 px).

How tight this pin can be is bounded by the reference itself, and the bounds below say so.  render.png shows the drone
with its albedo / EMISSION / metallic / roughness / normal maps bound: a glowing yellow ring and blue lamps that light the
floor and everything near them.  Those five TGA files are missing from the reference tree (.MISSING_LARGE_BLOBS), so
Texture::load_from_file returns None for them today (texture.rs:22-24) and the rebuilt scene has a black, unlit drone.
Consequences, all visible in the numbers: (i) every region is at most as bright here as in render.png (light is only
missing, never added) — a ONE-SIDED bound that holds everywhere; (ii) the sphere grid at z = 0, lit almost only by the
ceiling light, agrees to a few 1/255 steps, the metallic = 1 column-group best (|d| <= 0.03) because it mirrors the light
and the black void; (iii) floor, volumes, glass sphere and the two meshes sit in the drone's glow and are 0.05 - 0.12
darker.  A 10 % change of one BSDF constant moves a sphere region by about 0.01 - that is below what the unknown drone
illumination leaves undetermined, so this test cannot, and does not claim to, resolve it; the oracle's arithmetic is pinned
by the analytic KATs (tests/test_oracle_kat.py), this test pins the scene literal, camera, tone map and gross energy.
CPU (oracle) here; the same comparison for the HIP image is in the -m gpu test below."""
import os

import numpy as np
import pytest

from cs397raytracingsp22_amd import scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def region_means(u8):
    z = np.load(os.path.join(GOLD, "reference_render_regions.npz"), allow_pickle=False)
    size = u8.shape[0]
    assert u8.shape[:2] == (size, size) and size % 200 == 0
    full = np.kron(z["labels"], np.ones((size // 200, size // 200), np.uint8))
    img = u8.astype(np.float32) / 255.0
    return [str(n) for n in z["names"]], z["ref_mean"], np.array([img[full == r + 1].mean(axis=0) for r in range(len(z["names"]))])


def check_regions(u8):
    names, ref, mine = region_means(u8)
    d = mine - ref
    grid = [i for i, n in enumerate(names) if n.startswith("pm_sphere")]
    assert len(grid) == 15 and len(names) >= 21
    for i, n in enumerate(names):
        # (i) light is only missing (the drone's emission map), never added
        assert float(d[i].max()) <= 0.03, (n, d[i])
        if n.startswith("pm_sphere") and n.endswith("metal1"):
            assert float(np.abs(d[i]).max()) <= 0.03, (n, d[i])           # (ii) mirrors: light + void, little drone glow
        elif n.startswith("pm_sphere"):
            assert float(d[i].min()) >= -0.06, (n, d[i])
        elif n == "cyan_emitter":
            # emission (0,1,1) saturates green and blue; the excess is pushed into red by the "saturate toward white" step
            # (tracing.rs:244-251) the same way on both sides
            assert mine[i][1] > 0.99 and mine[i][2] > 0.99 and ref[i][1] > 0.99 and abs(float(d[i][0])) <= 0.05, (n, mine[i], ref[i])
        else:
            assert float(d[i].min()) >= -0.14, (n, d[i])                  # (iii) in the drone's glow
    assert float(np.abs(d[grid]).mean()) <= 0.025, float(np.abs(d[grid]).mean())
    # hue and ordering survive the missing glow: the grid is blue, the mesh sphere magenta, the cube green, the emitter cyan
    for i, n in enumerate(names):
        if n.startswith("pm_sphere"):
            assert mine[i][2] > mine[i][0] and mine[i][2] > mine[i][1]
    ms, cb = names.index("sphere_mesh"), names.index("cube_mesh")
    assert mine[ms][1] < 0.03 and mine[ms][0] > 0.15 and mine[ms][2] > 0.15 and mine[cb][1] > 1.5 * mine[cb][0]
    # roughness ladder of the dielectric-like row (metallic 0): the sharper the lobe, the darker the red/green channels get
    row0 = [names.index(f"pm_sphere_rough{r}_metal0") for r in ("0.25", "0.5", "0.75", "1")]
    assert mine[row0[0]][0] >= mine[row0[-1]][0] and ref[row0[0]][0] >= ref[row0[-1]][0]
    return names, ref, mine


def test_oracle_matches_published_render_region_by_region(orc):
    sc = scenes.head_scene(200, 200, 36, 10, textures=scenes.load_asset_textures())
    _, u8, _, _ = orc.OracleScene(sc.flatten()).render(sc.camera, seed=1, want_sig=False)
    check_regions(u8)
    # the coarse check of round 1 still holds: block means outside the drone
    ref50 = np.load(os.path.join(GOLD, "reference_render_50x50.npy"))
    mine50 = (u8.astype(np.float32) / 255.0).reshape(50, 4, 50, 4, 3).mean(axis=(1, 3))
    yy, xx = np.mgrid[0:50, 0:50] / 50.0
    drone = (xx > 0.18) & (xx < 0.78) & (yy > 0.40) & (yy < 0.92)
    assert float(np.abs(mine50 - ref50)[~drone].mean()) < 0.035


@pytest.mark.gpu
def test_hip_image_matches_published_render_region_by_region(gpu_ctx):
    """The same comparison for the image the HIP path produces, at render.png's own 800x800 and 256 spp."""
    sc = scenes.head_scene(800, 800, 256, 10, textures=scenes.load_asset_textures())
    gpu_ctx.upload(sc.flatten())
    _, u8, _, _ = gpu_ctx.render(sc.camera, seed=1, want_f32=False)
    names, ref, mine = check_regions(u8)
    for n, r, m in zip(names, ref, mine):
        print(f"{n:34s} render.png {np.round(r, 3)}  hip {np.round(m, 3)}")
