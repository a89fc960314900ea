"""Edge cases of the hot path, HIP (through the C ABI) against the oracle — same bars as tests/test_gpu_parity.py.
The reference holds no tests; these are the boundaries its code has: an empty Scene.objects (every ray is background,
tracing.rs:306), path_depth 0 (shade_ray returns the background before intersecting, tracing.rs:301), a 1x1 image, one
sample per pixel and sample counts that are no perfect square (strata from floor(sqrt(n)), tracing.rs:165-170), images thinner
than one 32-px tile, meshes whose root is a leaf (geometry.rs:95) or that consist of two triangles, a scene with nothing but
a mesh, more than 64 list entries (tile masks switch themselves off), a tiny max_trace_dist, and a camera inside an object."""
import numpy as np
import pytest

from cs397raytracingsp22_amd import (Camera, ConvexVolume, Dielectric, Isotropic, Lambertian, Metal, Plane, Scene, Sphere, StaticMesh,
                                     Triangle, abi, cgmath, scenes)
from cs397raytracingsp22_amd.objload import Mesh

from test_gpu_parity import compare

pytestmark = pytest.mark.gpu

VARIANTS = [abi.MI_VARIANT_DEFAULT, abi.MI_VARIANT_VOTED, abi.MI_VARIANT_RECURSIVE]


def camera(w, h, spp, depth, **kw):
    args = dict(eyepoint=(0.0, 2.5, 7.5), view_dir=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), path_depth=depth, path_samples=1,
                screen_width=w, screen_height=h, focal_length=0.8, focus_dist=5.0, lens_radius=0.0, aa_sample_count=spp,
                max_trace_dist=100.0, gamma=2.0)
    args.update(kw)
    return Camera(**args)


def tiny_mesh(n_tris):
    """n_tris triangles in a fan around the y axis, one vertex set per triangle (tobj single_index layout)."""
    pos, nrm, uv, idx = [], [], [], []
    for k in range(n_tris):
        a0, a1 = 2.0 * np.pi * k / max(n_tris, 3), 2.0 * np.pi * (k + 1) / max(n_tris, 3)
        tri = [(0.0, 1.0, 0.0), (np.cos(a0), 0.0, np.sin(a0)), (np.cos(a1), 0.0, np.sin(a1))]
        n = np.cross(np.subtract(tri[1], tri[0]), np.subtract(tri[2], tri[0]))
        n = n / np.linalg.norm(n)
        for j, p in enumerate(tri):
            pos.append(p); nrm.append(tuple(n)); uv.append(((j == 1) * 1.0, (j == 2) * 1.0)); idx.append(3 * k + j)
    return Mesh(np.asarray(pos, np.float32), np.asarray(nrm, np.float32), np.asarray(uv, np.float32), np.asarray(idx, np.uint32), f"fan{n_tris}")


@pytest.mark.parametrize("variant", VARIANTS)
def test_empty_scene_is_background(gpu_ctx, orc, variant):
    st = compare(gpu_ctx, orc, Scene(camera(70, 45, 4, 5), []), variant=variant)
    assert st.samples == 70 * 45 * 4


@pytest.mark.parametrize("variant", VARIANTS)
def test_path_depth_zero_never_intersects(gpu_ctx, orc, variant):
    sc = scenes.config2(64, 48, 4, 0)
    compare(gpu_ctx, orc, sc, variant=variant)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("w,h,spp,depth", [(1, 1, 1, 1), (1, 1, 7, 3), (33, 1, 2, 4), (1, 33, 3, 4), (31, 31, 5, 2), (65, 2, 10, 6)])
def test_tiny_images_and_odd_sample_counts(gpu_ctx, orc, variant, w, h, spp, depth):
    compare(gpu_ctx, orc, scenes.config2(w, h, spp, depth), variant=variant, seed=w * 131 + h)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("n_tris", [1, 2, 3])
def test_mesh_of_one_two_three_triangles(gpu_ctx, orc, variant, n_tris):
    """One triangle: the root IS a leaf and is never box-tested (geometry.rs:95); two: the smallest interior node."""
    xf = cgmath.mul(cgmath.from_translation((0.0, 1.5, 0.0)), cgmath.from_angle_x(35.0), cgmath.from_scale(2.0))
    mesh = StaticMesh(tiny_mesh(n_tris), Lambertian(albedo=(0.8, 0.3, 0.2), emission=(0.1, 0.1, 0.1)), [None] * 5, xf)
    sc = Scene(camera(80, 60, 4, 4), scenes.cornell_walls() + [mesh])
    compare(gpu_ctx, orc, sc, variant=variant)


@pytest.mark.parametrize("variant", VARIANTS)
def test_scene_with_nothing_but_a_mesh(gpu_ctx, orc, variant):
    xf = cgmath.mul(cgmath.from_translation((0.0, 1.5, 0.0)), cgmath.from_scale(1.5))
    sc = Scene(camera(90, 64, 4, 6), [StaticMesh(scenes.load_asset_mesh("teapot"), Metal(albedo=(0.9, 0.8, 0.5), emission=(0.2, 0.2, 0.2), roughness=0.2),
                                                  [None] * 5, xf)])
    compare(gpu_ctx, orc, sc, variant=variant)


@pytest.mark.parametrize("variant", [abi.MI_VARIANT_DEFAULT, abi.MI_VARIANT_VOTED])
def test_more_than_64_list_entries(gpu_ctx, orc, variant):
    """A 64-bit tile mask cannot describe 70 + 10 entries: masking must switch itself off, not truncate."""
    rng = np.random.default_rng(5)
    objs = scenes.cornell_walls()
    for _ in range(70):
        c = rng.uniform((-2.3, 0.3, -2.3), (2.3, 4.5, 2.3))
        objs.append(Sphere(tuple(map(float, c)), float(rng.uniform(0.08, 0.25)), Lambertian(albedo=tuple(map(float, rng.uniform(0.2, 0.9, 3))), emission=(0, 0, 0))))
    f = scenes.config1(100, 80, 4, 5)
    sc = Scene(f.camera, objs)
    flat = sc.flatten()
    assert flat.desc.n_objects == len(objs) > 64
    compare(gpu_ctx, orc, sc, variant=variant)
    # and the masked / unmasked renders agree bit for bit
    gpu_ctx.upload(flat)
    _, _, s0, _ = gpu_ctx.render(sc.camera, seed=3, want_u8=False, want_sig=True)
    _, _, s1, _ = gpu_ctx.render(sc.camera, seed=3, want_u8=False, want_sig=True, flags=abi.MI_OPT_NO_TILE_MASKS)
    assert np.array_equal(s0, s1)


@pytest.mark.parametrize("variant", VARIANTS)
def test_spheres_that_give_a_ray_many_candidates(gpu_ctx, orc, variant):
    """The list's Sphere tests run in two stages (pt_kernels.hip sphere_stage1 / sphere_finish: the discriminant for every
    sphere, the roots only for the lanes that passed it, from a one-entry stash that is flushed whenever a lane needs it
    again).  This scene makes a ray pass the discriminant of MANY spheres in a row — concentric glass shells around the eye's
    line of sight, a cluster of overlapping spheres, the camera inside three of them — and lists the same sphere twice with
    two materials (equal distances: the first Scene.objects entry wins, tracing.rs:335), once directly behind each other and
    once with other spheres in between."""
    twin = ((0.9, 1.2, 0.5), 0.55)
    objs = scenes.cornell_walls()
    objs.append(Sphere(*twin, Lambertian(albedo=(0.9, 0.1, 0.1), emission=(0.0, 0.0, 0.0))))
    objs.append(Sphere(*twin, Lambertian(albedo=(0.1, 0.9, 0.1), emission=(2.0, 2.0, 2.0))))          # same sphere again: never seen
    for r in (0.25, 0.5, 0.75, 1.0, 1.25):                                                             # concentric shells
        objs.append(Sphere((-0.8, 2.2, 0.0), r, Dielectric(idx_of_refraction=1.0 + 0.1 * r)))
    rng = np.random.default_rng(17)
    for _ in range(24):                                                                                # an overlapping cluster
        c = np.array((1.2, 3.6, -0.6)) + rng.uniform(-0.5, 0.5, 3)
        mat = Metal(albedo=tuple(map(float, rng.uniform(0.3, 0.9, 3))), emission=(0, 0, 0), roughness=float(rng.uniform(0, 0.6))) \
            if rng.random() < 0.5 else Lambertian(albedo=tuple(map(float, rng.uniform(0.2, 0.9, 3))), emission=(0, 0, 0))
        objs.append(Sphere(tuple(map(float, c)), float(rng.uniform(0.3, 0.6)), mat))
    objs.append(Sphere(*twin, Metal(albedo=(0.1, 0.1, 0.9), emission=(3.0, 0.0, 0.0), roughness=0.0)))   # the twin a third time, far down the list
    for r in (0.6, 1.2, 2.4):                                                                          # the eye sits inside these
        objs.append(Sphere((0.0, 2.5, 7.5), r, Dielectric(idx_of_refraction=1.3)))
    compare(gpu_ctx, orc, Scene(camera(96, 72, 4, 8), objs), variant=variant)


@pytest.mark.parametrize("variant", VARIANTS)
def test_tiny_max_trace_dist_and_camera_inside_objects(gpu_ctx, orc, variant):
    """max_trace_dist shorter than the room (most rays end as background although geometry lies ahead), the eye inside a
    glass sphere and inside a ConvexVolume (t_entr < 0: geometry.rs:501-513)."""
    objs = scenes.cornell_walls() + [Sphere((0.0, 2.5, 7.5), 0.6, Dielectric(1.5)),
                                     ConvexVolume(Sphere((0.0, 2.5, 7.0), 2.0, Dielectric(1.5)), Isotropic(albedo=(0.9, 0.9, 0.9)), 0.8),
                                     Plane((0.0, -0.2, 0.0), (0.0, 1.0, 0.0), Lambertian(albedo=(0.4, 0.4, 0.4), emission=(0, 0, 0)))]
    for dist in (100.0, 6.0, 0.5, float("inf")):           # inf: "no limit" is a legitimate max_trace_dist
        compare(gpu_ctx, orc, Scene(camera(72, 54, 4, 6, max_trace_dist=dist), objs), variant=variant)


@pytest.mark.parametrize("variant", [abi.MI_VARIANT_DEFAULT, abi.MI_VARIANT_VOTED])
@pytest.mark.parametrize("flags", [0, abi.MI_OPT_TWO_STAGE])
def test_more_than_32_meshes(gpu_ctx, orc, variant, flags):
    """The per-tile mesh masks and the walkers' mesh masks are 32 bits wide: meshes 32, 33, ... must still be walked (and
    their trees must be inside the walker's LDS window), whatever the masks say about the first 32."""
    rng = np.random.default_rng(11)
    objs = scenes.cornell_walls()
    cube, teapot = scenes.load_asset_mesh("cube"), scenes.load_asset_mesh("teapot")
    for k in range(37):
        c = rng.uniform((-2.4, 0.4, -2.4), (2.4, 5.0, 2.0))
        xf = cgmath.mul(cgmath.from_translation(tuple(map(float, c))), cgmath.from_angle_y(float(rng.uniform(0, 360))),
                        cgmath.from_angle_x(float(rng.uniform(0, 90))), cgmath.from_scale(float(rng.uniform(0.15, 0.4))))
        objs.append(StaticMesh(teapot if k in (5, 34) else cube, Lambertian(albedo=tuple(map(float, rng.uniform(0.2, 0.9, 3))), emission=(0, 0, 0)), [None] * 5, xf))
    sc = Scene(scenes.config1(120, 90, 4, 6).camera, objs)
    flat = sc.flatten()
    assert flat.desc.n_meshes == 37
    gpu_ctx.upload(flat)
    f32, u8, sig, _ = gpu_ctx.render(sc.camera, seed=2, want_sig=True, variant=variant, flags=flags)
    r32, r8, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=2)
    assert int((sig != rsig).sum()) == 0
    assert float(np.abs(f32 - r32).max()) <= 2e-5 * max(1.0, float(np.abs(r32).max()))
    assert int(np.abs(u8.astype(int) - r8.astype(int)).max()) <= 1


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("case", ["degenerate_objects", "zero_mesh_normals"])
def test_degenerate_geometry(gpu_ctx, orc, variant, case):
    """Garbage in, the reference's garbage out: zero-area triangles, a zero-radius sphere and zero vertex normals
    (normalize(0) = NaN normal, geometry.rs:356 -> NaN scatter directions) walk through the same decisions as the oracle."""
    objs = scenes.config2(8, 8, 1, 1).objects
    if case == "degenerate_objects":
        objs = objs + [Triangle((0, 1, 0), (0, 1, 0), (0, 1, 0), Lambertian(albedo=(0.5, 0.5, 0.5), emission=(1, 1, 1))),
                       Triangle((-1, 1, 0), (0, 1, 0), (1, 1, 0), Lambertian(albedo=(0.5, 0.5, 0.5), emission=(1, 1, 1))),
                       Sphere((0.5, 2.0, 1.0), 0.0, Metal(albedo=(0.9, 0.9, 0.9), emission=(0, 0, 0), roughness=0.0))]
    else:
        m = tiny_mesh(3)
        m.normals[:] = 0.0
        xf = cgmath.mul(cgmath.from_translation((0.0, 1.0, 1.0)), cgmath.from_scale(2.0))
        objs = objs[:12] + [StaticMesh(m, Lambertian(albedo=(0.7, 0.7, 0.7), emission=(0.2, 0.2, 0.2)), [None] * 5, xf)]
    sc = Scene(camera(48, 40, 4, 5), objs)
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    f32, u8, sig, _ = gpu_ctx.render(sc.camera, seed=4, want_sig=True, variant=variant)
    r32, r8, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=4)
    assert int((sig != rsig).sum()) == 0
    fin = np.isfinite(r32)
    assert np.array_equal(np.isfinite(f32), fin)
    assert float(np.abs(f32[fin] - r32[fin]).max(initial=0.0)) <= 2e-5 * max(1.0, float(np.abs(r32[fin]).max(initial=0.0)))
    assert int(np.abs(u8.astype(int) - r8.astype(int)).max()) <= 1


@pytest.mark.parametrize("field,value", [("up", (0.0, 0.0, -2.0)), ("view_dir", (0.0, 0.0, 0.0)), ("eyepoint", (0.0, float("nan"), 0.0)),
                                         ("up", (0.0, float("inf"), 0.0)), ("focus_dist", float("nan")), ("max_trace_dist", float("nan"))])
def test_cameras_that_make_every_ray_non_finite_are_refused(gpu_ctx, field, value):
    """A DEVIATION from the reference, on purpose: the reference renders such a camera — every ray is NaN,
    its tests 'hit' with NaN distances and Scene keeps the first such hit in Scene.objects order (tracing.rs:335), which the
    kind-grouped object list of the kernels does not reproduce.  The library says so instead of returning a different image."""
    sc = scenes.config2(32, 32, 1, 2)
    gpu_ctx.upload(sc.flatten())
    cam = camera(32, 32, 1, 2, **{field: value})
    with pytest.raises(abi.MiError) as ei:
        gpu_ctx.render(cam)
    assert ei.value.code == abi.MI_ERR_INVALID
    f32, _, _, _ = gpu_ctx.render(camera(32, 32, 1, 2))          # the context is still usable
    assert np.isfinite(f32).all()


def test_more_ranks_than_tiles(gpu_ctx):
    """A 40x20 image is two tiles: with world 5, ranks 2, 3, 4 own nothing.  Every variant must cope with an empty share
    (and still write its padding slot as zeros), and the assembly equals the one-rank image bit for bit."""
    from test_gpu_tiles import assemble
    sc = scenes.config2(40, 20, 4, 5)
    gpu_ctx.upload(sc.flatten())
    ref32, ref8, _, _ = gpu_ctx.render(sc.camera, seed=3)
    for world in (2, 5, 8):
        img, u8, gathered = assemble(gpu_ctx, sc.camera, world)
        assert np.array_equal(img, ref32) and np.array_equal(u8, ref8), world
        assert not np.isnan(gathered).any()


@pytest.mark.parametrize("variant", [abi.MI_VARIANT_DEFAULT, abi.MI_VARIANT_VOTED, abi.MI_VARIANT_SIMPLE])
def test_hundreds_of_list_objects_of_every_kind(gpu_ctx, orc, variant):
    rng = np.random.default_rng(21)
    objs = scenes.cornell_walls()
    for k in range(420):
        c = rng.uniform((-2.6, 0.2, -2.6), (2.6, 5.6, 2.6))
        mat = Lambertian(albedo=tuple(map(float, rng.uniform(0.2, 0.9, 3))), emission=(0, 0, 0)) if k % 3 else \
            Metal(albedo=(0.8, 0.8, 0.8), emission=(0, 0, 0), roughness=float(rng.uniform(0, 0.5)))
        if k % 4 == 0:
            p = c + rng.uniform(-0.3, 0.3, (3, 3))
            objs.append(Triangle(tuple(map(float, p[0])), tuple(map(float, p[1])), tuple(map(float, p[2])), mat))
        elif k % 41 == 0:
            objs.append(ConvexVolume(Sphere(tuple(map(float, c)), 0.3, Dielectric(1.5)), Isotropic(albedo=(0.8, 0.8, 0.8)), 2.0))
        elif k % 97 == 0:
            objs.append(Plane((0.0, float(-0.1 - 0.01 * k), 0.0), (0.0, 1.0, 0.0), mat))
        else:
            objs.append(Sphere(tuple(map(float, c)), float(rng.uniform(0.05, 0.2)), mat))
    objs = [objs[i] for i in rng.permutation(len(objs))]
    compare(gpu_ctx, orc, Scene(scenes.config1(96, 72, 4, 5).camera, objs), variant=variant)


@pytest.mark.parametrize("flags", [0, abi.MI_OPT_TWO_STAGE])
@pytest.mark.parametrize("name", ["teapot", "sphere"])
def test_sheared_mirrored_non_uniform_mesh_transforms(gpu_ctx, orc, name, flags):
    """StaticMesh transforms the reference allows but its own scenes never use: non-uniform scale, shear, a mirror
    (negative determinant).  Normals go through (M^-1)^T (geometry.rs:297); the two-stage bound must hold for the longer
    object-space directions such a matrix produces — or the mesh must fall back to the reference walk."""
    mesh = scenes.load_asset_mesh(name)
    base = {"teapot": 1.2, "sphere": 1.0}[name]
    shear = np.eye(4, dtype=np.float32)
    shear[0, 1] = 0.6; shear[2, 0] = -0.4
    for k, S in enumerate(((2.5, 0.3, 1.0), (-1.0, 1.0, 1.0), (0.2, 1.8, -0.7))):
        scale = np.diag([S[0] * base, S[1] * base, S[2] * base, 1.0]).astype(np.float32)
        xf = cgmath.mul(cgmath.from_translation((0.0, 2.0, 0.0)), cgmath.from_angle_y(30.0 * k), shear if k == 2 else np.eye(4, dtype=np.float32), scale)
        objs = scenes.cornell_walls() + [StaticMesh(mesh, Metal(albedo=(0.9, 0.7, 0.4), emission=(0.05, 0.05, 0.05), roughness=0.3), [None] * 5, xf)]
        sc = Scene(camera(64, 48, 4, 5), objs)
        flat = sc.flatten()
        gpu_ctx.upload(flat)
        f32, u8, sig, _ = gpu_ctx.render(sc.camera, seed=9, want_sig=True, flags=flags)
        r32, r8, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=9)
        assert int((sig != rsig).sum()) == 0, (name, k, flags)
        assert float(np.abs(f32 - r32).max()) <= 2e-5 * max(1.0, float(np.abs(r32).max()))
        assert int(np.abs(u8.astype(int) - r8.astype(int)).max()) <= 1


@pytest.mark.parametrize("variant", [abi.MI_VARIANT_DEFAULT, abi.MI_VARIANT_VOTED, abi.MI_VARIANT_RECURSIVE])
def test_textures_of_odd_sizes(gpu_ctx, orc, variant):
    """Texture::sample (texture.rs:26-32) on 1x1, 1xN, Nx1 and mutually different map sizes (the per-mesh interleaved texel
    record needs equal sizes; these take the five-fetch path), and on equal non-square sizes (interleaved)."""
    from cs397raytracingsp22_amd import Texture
    rng = np.random.default_rng(3)
    tex = lambda w, h: Texture(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    mesh = scenes.load_asset_mesh("sphere")           # texcoords beyond [0, 1]: clamped, not wrapped
    for maps in ([tex(1, 1), tex(1, 7), tex(5, 1), tex(3, 2), tex(16, 9)], [tex(7, 3)] * 5, [tex(1, 1), None, None, None, None],
                 [tex(4, 4), None, tex(2, 8), None, tex(4, 4)]):
        xf = cgmath.mul(cgmath.from_translation((0.0, 2.0, 0.0)), cgmath.from_scale(1.6))
        sc = Scene(camera(56, 44, 4, 4), scenes.cornell_walls() + [StaticMesh(mesh, None, maps, xf)])
        compare(gpu_ctx, orc, sc, variant=variant, seed=6)


def _many_triangles(rng, n, lo=(-2.6, 0.2, -2.6), hi=(2.6, 5.6, 2.6), size=0.35):
    objs = []
    for k in range(n):
        c = rng.uniform(lo, hi)
        p = c + rng.uniform(-size, size, (3, 3))
        mat = Lambertian(albedo=tuple(map(float, rng.uniform(0.2, 0.9, 3))), emission=(0.3, 0.3, 0.3) if k % 7 == 0 else (0, 0, 0)) if k % 3 else \
            Metal(albedo=(0.8, 0.8, 0.8), emission=(0, 0, 0), roughness=float(rng.uniform(0, 0.5)))
        objs.append(Triangle(tuple(map(float, p[0])), tuple(map(float, p[1])), tuple(map(float, p[2])), mat))
    return objs


@pytest.mark.parametrize("variant", [abi.MI_VARIANT_DEFAULT, abi.MI_VARIANT_VOTED])
def test_long_triangle_lists_walk_a_top_level_tree(gpu_ctx, orc, variant):
    """96 or more small Triangles in Scene.objects: the scene compiler puts them into a top-level tree (the walls — 32 x the median
    |e1||e2| and more — stay in front and are tested one by one) and the wavefront pipeline's object list walks it with padded boxes,
    running the reference's own test on the leaves it reaches.  Same paths as the oracle's linear loop, and as the library's own with
    MI_OPT_NO_LIST_TREE; a glass sphere and a metal one keep the bounce directions off unit length."""
    rng = np.random.default_rng(77)
    objs = scenes.cornell_walls() + _many_triangles(rng, 150) + scenes.cornell_spheres()
    # degenerate and awkward members of the tree: a zero-area triangle, a sliver, two identical triangles with different materials
    objs.append(Triangle((0.5, 1.0, 0.5), (0.5, 1.0, 0.5), (0.7, 1.3, 0.5), Lambertian(albedo=(0.9, 0.9, 0.1))))
    objs.append(Triangle((-1.0, 2.0, 0.0), (1.0, 2.0, 1e-6), (0.0, 2.0, 5e-7), Lambertian(albedo=(0.1, 0.9, 0.9))))
    twin = ((-0.6, 3.0, 0.2), (0.1, 3.1, 0.3), (-0.2, 3.6, 0.1))
    objs.append(Triangle(*twin, Lambertian(albedo=(0.9, 0.1, 0.1))))
    objs.append(Triangle(*twin, Lambertian(albedo=(0.1, 0.9, 0.1), emission=(2.0, 2.0, 2.0))))     # never seen: the first entry wins the tie
    objs = [objs[i] for i in rng.permutation(len(objs))]
    sc = Scene(camera(110, 80, 4, 7), objs)
    compare(gpu_ctx, orc, sc, variant=variant)
    if variant == abi.MI_VARIANT_DEFAULT:
        flat = sc.flatten()
        gpu_ctx.upload(flat)
        f0, _, s0, _ = gpu_ctx.render(sc.camera, seed=3, want_u8=False, want_sig=True)
        f1, _, s1, _ = gpu_ctx.render(sc.camera, seed=3, want_u8=False, want_sig=True, flags=abi.MI_OPT_NO_LIST_TREE)
        assert np.array_equal(s0, s1) and np.array_equal(f0, f1)
        gpu_ctx.render(sc.camera, seed=3, want_u8=False)
        with_tree = gpu_ctx.last_pipeline_ms()["wf_main_ms"]
        gpu_ctx.render(sc.camera, seed=3, want_u8=False, flags=abi.MI_OPT_NO_LIST_TREE)
        assert gpu_ctx.last_pipeline_counts()["segments"] > 0 and with_tree > 0


def test_top_level_tree_rays_the_bound_does_not_cover(gpu_ctx, orc):
    """The padding bound needs B = 7 eps E2 |d| / 1e-4 <= 1/2.  A hundred and twenty triangles of |e1||e2| ~ 50 (tree built: B = 0.21 |d|) under an
    orthographic camera whose view_dir has length 3 (tracing.rs:200: the direction is not normalised) send every camera ray in
    with B = 0.63: those waves test the list one by one; the bounce rays (in-ball directions) walk the tree.  Same paths as the oracle."""
    rng = np.random.default_rng(5)
    objs = []
    for k in range(120):
        c = rng.uniform((-12.0, -12.0, -30.0), (12.0, 12.0, -10.0))
        p = c + rng.uniform(-5.0, 5.0, (3, 3))
        objs.append(Triangle(tuple(map(float, p[0])), tuple(map(float, p[1])), tuple(map(float, p[2])),
                             Lambertian(albedo=tuple(map(float, rng.uniform(0.3, 0.9, 3))), emission=(0.2, 0.2, 0.2))))
    objs.append(Sphere((0.0, 0.0, -20.0), 3.0, Metal(albedo=(0.9, 0.9, 0.9), emission=(0, 0, 0), roughness=0.05)))
    cam = camera(96, 72, 4, 5, projection_mode=abi.MI_PROJ_ORTHOGRAPHIC, eyepoint=(0.0, 0.0, 8.0), view_dir=(0.0, 0.0, -3.0), focal_length=0.05)
    compare(gpu_ctx, orc, Scene(cam, objs))


def test_long_lists_with_triangles_of_one_size_everywhere(gpu_ctx, orc):
    """No outliers: every triangle goes into the tree (nothing stays in front), lists of exactly the minimum length, and a list one short
    of it (no tree at all)."""
    for n in (96, 95, 300):
        rng = np.random.default_rng(100 + n)
        objs = _many_triangles(rng, n, size=0.5) + [Sphere((0.0, 2.5, 0.0), 0.8, Dielectric(1.5))]
        compare(gpu_ctx, orc, Scene(camera(80, 60, 4, 6), objs), seed=n)
