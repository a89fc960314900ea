"""Primary-ray tile masks (mi_rt.cpp tile_masks): the wavefront pipeline skips, for the camera rays of a
32x32 tile, the list entries a conservative host-side frustum test proves unreachable.  Skipping must
never change a path: random scenes of many small objects under random (also skewed, non-unit) camera
bases, checked bit for bit against the oracle and against the same render with masking switched off."""
import numpy as np
import pytest

from cs397raytracingsp22_amd import abi  # noqa: E402
from cs397raytracingsp22_amd import (Camera, ConvexVolume, Dielectric, Isotropic, Lambertian, Metal, Plane, Scene, Sphere,
                                     StaticMesh, Triangle, cgmath, scenes)

pytestmark = pytest.mark.gpu


def scatter_scene(seed, n_tri=40, n_sph=12, skew=True):
    rng = np.random.default_rng(seed)
    mats = [Lambertian(albedo=(0.7, 0.7, 0.7), emission=(0.5, 0.5, 0.5)), Metal(albedo=(0.8, 0.6, 0.4), roughness=0.2),
            Dielectric(idx_of_refraction=1.5), Lambertian(albedo=(0.2, 0.6, 0.3))]
    eye = rng.uniform(-1, 1, 3)
    view = rng.normal(size=3); view /= np.linalg.norm(view)
    view *= rng.uniform(0.7, 1.4) if skew else 1.0                      # the reference never normalises view_dir
    up = rng.normal(size=3)
    up -= (0.0 if skew else 1.0) * np.dot(up, view) * view / np.dot(view, view)   # skew: NOT orthogonalised
    up /= np.linalg.norm(up)
    up *= rng.uniform(0.8, 1.3) if skew else 1.0
    objs = []
    for _ in range(n_tri):                                                # all around the eye, also behind it
        c = eye + rng.normal(size=3) * 4.0
        a, b, cc = (c + rng.normal(size=3) * rng.uniform(0.05, 0.8) for _ in range(3))
        objs.append(Triangle(a=tuple(a), b=tuple(b), c=tuple(cc), material=mats[int(rng.integers(len(mats)))]))
    for i in range(n_sph):
        c = eye + rng.normal(size=3) * 4.0
        r = rng.uniform(0.05, 0.7) if i else 1.6                          # sphere 0 is big, sometimes around the eye
        if i == 0 and seed % 3 == 0:
            c = eye + rng.normal(size=3) * 0.3
        objs.append(Sphere(center=tuple(c), radius=float(r), material=mats[int(rng.integers(len(mats)))]))
    order = rng.permutation(len(objs))
    cam = Camera(eyepoint=tuple(eye), view_dir=tuple(view), up=tuple(up), path_depth=4, path_samples=1,
                 screen_width=int(rng.integers(100, 260)), screen_height=int(rng.integers(70, 150)),
                 focal_length=float(rng.uniform(0.3, 1.2)), focus_dist=float(rng.uniform(2, 8)), lens_radius=0.0,
                 aa_sample_count=int(rng.choice([1, 4, 9])), max_trace_dist=100.0, gamma=2.0)
    return Scene(cam, [objs[i] for i in order])


@pytest.mark.parametrize("seed", list(range(16)))
def test_masked_primary_rays_take_the_oracles_paths(gpu_ctx, orc, seed):
    sc = scatter_scene(500 + seed, skew=(seed % 2 == 0))
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    f32, _, sig, _ = gpu_ctx.render(sc.camera, seed=seed, want_u8=False, want_sig=True)
    r32, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=seed, want_u8=False)
    assert int((sig != rsig).sum()) == 0
    assert float((np.abs(f32.astype(np.float64) - r32) / np.maximum(1.0, np.abs(r32))).max()) <= 2e-5
    g32, _, gsig, _ = gpu_ctx.render(sc.camera, seed=seed, want_u8=False, want_sig=True, flags=abi.MI_OPT_NO_TILE_MASKS)
    assert np.array_equal(gsig, sig) and np.array_equal(g32, f32)          # masking changes nothing, bit for bit


def test_more_than_64_entries_disables_masking(gpu_ctx, orc):
    sc = scatter_scene(77, n_tri=70, n_sph=6)
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    _, _, sig, _ = gpu_ctx.render(sc.camera, seed=3, want_u8=False, want_sig=True)
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=3, want_u8=False)
    assert np.array_equal(sig, rsig)


def mesh_scene(seed, extras):
    """Cubes and teapots under random affine transforms (rotation, non-uniform scale) around a random camera;
    `extras` adds the never-masked kinds, which also keep every tile alive."""
    rng = np.random.default_rng(seed)
    sc = scatter_scene(seed, n_tri=10, n_sph=4, skew=(seed % 2 == 1))
    eye = np.asarray(sc.camera.eyepoint)
    objs = list(sc.objects)
    for k in range(4):
        name = "cube" if k % 2 == 0 else "teapot"
        xf = cgmath.mul(cgmath.from_translation(tuple(eye + rng.normal(size=3) * 3.5)),
                        cgmath.from_angle_y(float(rng.uniform(0, 360))), cgmath.from_angle_x(float(rng.uniform(0, 360))),
                        cgmath.from_nonuniform_scale(*[float(v) for v in rng.uniform(0.2, 1.0, 3)]))
        objs.append(StaticMesh(scenes.load_asset_mesh(name), Lambertian(albedo=(0.6, 0.3, 0.2), emission=(0.2, 0.2, 0.2)), [None] * 5, xf))
    if extras:
        objs.append(Plane(point=tuple(eye + np.float32([0, -3, 0])), normal=(0.0, 1.0, 0.0), material=Lambertian(albedo=(0.5, 0.5, 0.5))))
        objs.append(ConvexVolume(boundary=Sphere(center=tuple(eye + rng.normal(size=3) * 2), radius=1.0, material=Lambertian()),
                                 density=1.5, phase_function=Isotropic(albedo=(0.9, 0.9, 0.9))))
    order = rng.permutation(len(objs))
    return Scene(sc.camera, [objs[i] for i in order])


@pytest.mark.parametrize("seed,extras", [(s, s % 3 == 0) for s in range(12)])
def test_masked_mesh_roots_and_dead_tiles(gpu_ctx, orc, seed, extras):
    sc = mesh_scene(900 + seed, extras)
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    f32, _, sig, _ = gpu_ctx.render(sc.camera, seed=seed, want_u8=False, want_sig=True)
    r32, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=seed, want_u8=False)
    assert int((sig != rsig).sum()) == 0
    assert float((np.abs(f32.astype(np.float64) - r32) / np.maximum(1.0, np.abs(r32))).max()) <= 2e-5
    # without signatures the dead-tile shortcut is live (no ray is generated for a tile that sees nothing)
    g32, _, _, _ = gpu_ctx.render(sc.camera, seed=seed, want_u8=False, want_sig=False)
    assert np.array_equal(g32, f32)


@pytest.mark.parametrize("height", [0.0, 1e-6, 1e-4, 1e-2])
def test_eye_in_the_plane_of_a_large_triangle(gpu_ctx, orc, height):
    """Grazing camera rays make Moller-Trumbore ill-conditioned (u, v of rays nearly in the triangle's plane are
    rounding noise); such triangles must stay in every tile's mask (mi_rt.cpp tile_masks, guard G)."""
    grey = Lambertian(albedo=(0.7, 0.7, 0.7), emission=(0.3, 0.3, 0.3))
    objs = [Triangle(a=(-40.0, 0.0, -60.0), b=(40.0, 0.0, -60.0), c=(0.0, 0.0, 5.0), material=grey),
            Triangle(a=(3.0, 0.0, -20.0), b=(9.0, 0.0, -20.0), c=(6.0, 0.0, -2.0), material=grey),       # off to the side
            Sphere(center=(0.0, 1.0, -8.0), radius=1.0, material=Metal(albedo=(0.8, 0.8, 0.8), roughness=0.1))]
    cam = Camera(eyepoint=(0.0, height, 2.0), view_dir=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), path_depth=3, path_samples=1,
                 screen_width=256, screen_height=96, focal_length=0.6, focus_dist=5.0, lens_radius=0.0,
                 aa_sample_count=16, max_trace_dist=100.0, gamma=2.0)
    sc = Scene(cam, objs)
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    _, _, sig, _ = gpu_ctx.render(cam, seed=5, want_u8=False, want_sig=True)
    _, _, rsig, _ = orc.OracleScene(flat).render(cam, seed=5, want_u8=False)
    assert np.array_equal(sig, rsig)


@pytest.mark.parametrize("name", ["config2", "config4", "config5"])
def test_every_tile_of_the_full_frames(gpu_ctx, name):
    """All 2040 tiles of the 1080p frames (the oracle windows of the other tests cover a handful): the masked
    render equals the unmasked one, signatures and radiance, bit for bit."""
    sc = {"config2": lambda: scenes.config2(1920, 1080, 16, 10), "config4": lambda: scenes.config4(1920, 1080, 4, 10, tex_size=256),
          "config5": lambda: scenes.config5(1920, 1080, 16, 50)}[name]()
    gpu_ctx.upload(sc.flatten())
    f32, _, sig, _ = gpu_ctx.render(sc.camera, seed=21, want_u8=False, want_sig=True)
    fast, _, _, _ = gpu_ctx.render(sc.camera, seed=21, want_u8=False, want_sig=False)     # dead-tile shortcut live
    g32, _, gsig, _ = gpu_ctx.render(sc.camera, seed=21, want_u8=False, want_sig=True, flags=abi.MI_OPT_NO_TILE_MASKS)
    assert np.array_equal(gsig, sig) and np.array_equal(g32, f32) and np.array_equal(fast, f32)
    assert int((sig != 0).sum()) > 0


def test_full_frame_against_the_recursive_kernel(gpu_ctx):
    """The same frame through two unrelated kernels — the masked wavefront pipeline and the one-lane-per-pixel
    recursive estimator (no masks, no compaction, no queues): every pixel of 1080p takes the same paths."""
    from cs397raytracingsp22_amd import abi
    sc = scenes.config2(1920, 1080, 4, 10)
    gpu_ctx.upload(sc.flatten())
    f32, _, sig, _ = gpu_ctx.render(sc.camera, seed=8, want_u8=False, want_sig=True)
    r32, _, rsig, _ = gpu_ctx.render(sc.camera, seed=8, want_u8=False, want_sig=True, variant=abi.MI_VARIANT_RECURSIVE)
    assert np.array_equal(sig, rsig)
    assert float((np.abs(f32.astype(np.float64) - r32) / np.maximum(1.0, np.abs(r32))).max()) <= 2e-5
