"""The exact two-stage mesh traversal on the GPU (wf_trav_f + wf_replay): every render must be
identical — signatures and radiance, bit for bit — to the same render walking the reference's own tree node by node
(MI_OPT_REFERENCE_WALK), and to the oracle.  Covers the default selection (large qualifying meshes), the forced mode
(every qualifying mesh, small ones included), several meshes per scene, the candidate-overflow path and rays the padding
bound does not cover."""
import numpy as np
import pytest

from cs397raytracingsp22_amd import (Camera, Lambertian, Metal, Scene, Sphere, StaticMesh, abi, cgmath, objload, scenes)

pytestmark = pytest.mark.gpu


def three_ways(ctx, sc, seed=1):
    ctx.upload(sc.flatten())
    out = {}
    for name, fl in (("default", 0), ("reference", abi.MI_OPT_REFERENCE_WALK), ("two_stage", abi.MI_OPT_TWO_STAGE)):
        f32, _, sig, st = ctx.render(sc.camera, seed=seed, want_u8=False, want_sig=True, flags=fl)
        out[name] = (f32, sig, st, ctx.last_pipeline_ms())
    ref32, refsig = out["reference"][0], out["reference"][1]
    for name in ("default", "two_stage"):
        assert np.array_equal(out[name][1], refsig), name
        assert np.array_equal(out[name][0], ref32), name
    return out


def test_head_scene_two_stage_equals_reference_walk_and_oracle(gpu_ctx, orc):
    """The reference's own run() scene: the 32512-triangle sphere.obj goes two-stage by default, drone and cube walk the
    reference's trees (the drone's object-space scale voids the bound, the cube is too small to qualify)."""
    sc = scenes.head_scene(320, 320, 16, 10, textures=scenes.load_asset_textures())
    out = three_ways(gpu_ctx, sc, seed=2)
    assert out["default"][3]["wf_trav_f_ms"] > 0.0 and out["default"][3]["wf_replay_ms"] > 0.0       # the new kernels ran
    assert out["reference"][3]["wf_trav_f_ms"] == 0.0
    win = (200, 230, 40, 24)                                   # on the sphere mesh (right foreground)
    x0, y0, w, h = win
    r32, _, rsig, _ = orc.OracleScene(sc.flatten()).render(sc.camera, seed=2, window=win, want_u8=False)
    f32, sig = out["default"][0], out["default"][1]
    assert np.array_equal(sig[y0:y0 + h, x0:x0 + w], rsig)
    assert float(np.sqrt(np.mean((f32[y0:y0 + h, x0:x0 + w].astype(np.float64) - r32) ** 2))) <= 1e-3
    print("HEAD 320x320x16: default %.2f ms, reference walk %.2f ms" % (out["default"][2].kernel_ms, out["reference"][2].kernel_ms))


def test_forced_two_stage_on_the_teapot(gpu_ctx, orc):
    """cfg2's 240-triangle teapot qualifies but is small (default: LDS walk of the reference's tree); forced through the
    two-stage kernels it must not change a bit."""
    sc = scenes.config2(480, 270, 16, 10)
    out = three_ways(gpu_ctx, sc, seed=5)
    assert out["two_stage"][3]["wf_trav_f_ms"] > 0.0 and out["default"][3]["wf_trav_f_ms"] == 0.0
    r32, _, rsig, _ = orc.OracleScene(sc.flatten()).render(sc.camera, seed=5, window=(200, 120, 64, 32), want_u8=False)
    assert np.array_equal(out["two_stage"][1][120:152, 200:264], rsig)


def _sphere_mesh():
    return scenes.load_asset_mesh("sphere")


def test_several_meshes_mixed_walks(gpu_ctx, orc):
    """Three sphere.obj instances (different scales: two qualify, the hugely scaled-down one has |d_obj| so large that the
    bound never covers it), the teapot and the drone in one scene, metal everywhere so that |d| grows along paths."""
    metal = Metal(albedo=(0.9, 0.9, 0.9), emission=(0.0, 0.0, 0.0), roughness=0.6)
    objs = scenes.cornell_walls()
    objs.append(StaticMesh(_sphere_mesh(), metal, [None] * 5, cgmath.mul(cgmath.from_translation((-1.5, 1.0, 0.0)), cgmath.from_scale(0.9))))
    objs.append(StaticMesh(_sphere_mesh(), Lambertian(albedo=(0.2, 0.6, 0.9)), [None] * 5,
                           cgmath.mul(cgmath.from_translation((1.4, 1.6, 0.5)), cgmath.from_angle_y(30.0), cgmath.from_scale(1.3))))
    objs.append(StaticMesh(_sphere_mesh(), metal, [None] * 5, cgmath.mul(cgmath.from_translation((0.0, 3.5, -1.0)), cgmath.from_scale(1e-4 * 6000))))
    objs.append(scenes.teapot_mesh())
    objs.append(Sphere((0.0, 0.6, 1.8), 0.5, metal))
    sc = Scene(scenes.cornell_camera(256, 192, 9, 12), objs)
    out = three_ways(gpu_ctx, sc, seed=9)
    win = (96, 64, 48, 32)
    x0, y0, w, h = win
    _, _, rsig, _ = orc.OracleScene(sc.flatten()).render(sc.camera, seed=9, window=win, want_u8=False)
    assert np.array_equal(out["default"][1][y0:y0 + h, x0:x0 + w], rsig)


def test_candidate_overflow_takes_the_reference_walk(gpu_ctx, orc):
    """Twelve coincident copies of one triangle plus a fan around a vertex: a ray through them passes more triangles than a
    queue entry has candidate slots, so that mesh is walked through the reference's tree for that ray — same result."""
    P, N, T, I = [], [], [], []
    for k in range(12):                                   # coincident stack
        b = len(P)
        P += [(-1.0, -1.0, 0.0), (1.0, -1.0, 0.0), (0.0, 1.2, 0.0)]
        I += [(b, b + 1, b + 2)]
    rng = np.random.default_rng(3)
    for k in range(1500):                                 # filler: small triangles so that the mesh is "large"
        c = rng.uniform(-2.0, 2.0, 3) + np.array([0.0, 0.0, -1.5])
        b = len(P)
        P += [tuple(c), tuple(c + rng.uniform(-0.1, 0.1, 3)), tuple(c + rng.uniform(-0.1, 0.1, 3))]
        I += [(b, b + 1, b + 2)]
    P = np.array(P, np.float32)
    N = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(P), 1))
    T = np.zeros((len(P), 2), np.float32)
    mesh = objload.Mesh(P, N, T, np.array(I, np.uint32), "stack")
    cam = Camera(eyepoint=(0.0, 0.0, 4.0), view_dir=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), path_depth=4, path_samples=1,
                 screen_width=96, screen_height=96, focal_length=0.6, focus_dist=5.0, lens_radius=0.0, aa_sample_count=4,
                 max_trace_dist=100.0, gamma=2.0)
    sc = Scene(cam, [StaticMesh(mesh, Lambertian(albedo=(0.7, 0.7, 0.7), emission=(0.5, 0.5, 0.5)), [None] * 5, cgmath.from_scale(1.0))])
    out = three_ways(gpu_ctx, sc, seed=4)
    _, _, rsig, _ = orc.OracleScene(sc.flatten()).render(sc.camera, seed=4, want_u8=False)
    assert np.array_equal(out["default"][1], rsig)
    assert int((rsig != 0).sum()) > 1000


def test_rays_the_bound_does_not_cover(gpu_ctx, orc):
    """A camera basis with |up| = 3000 makes primary rays with |d| in the thousands: B > 1/2 for those rays, which must take
    the reference walk inside wf_replay while later (in-ball Lambertian) segments of the same paths go two-stage."""
    objs = scenes.cornell_walls()
    objs.append(StaticMesh(_sphere_mesh(), Lambertian(albedo=(0.8, 0.3, 0.3)), [None] * 5,
                           cgmath.mul(cgmath.from_translation((0.0, 2.0, 0.0)), cgmath.from_scale(1.5))))
    cam = scenes.cornell_camera(160, 120, 4, 6)
    cam.up = (0.0, 3000.0, 0.0)
    sc = Scene(cam, objs)
    out = three_ways(gpu_ctx, sc, seed=6)
    _, _, rsig, _ = orc.OracleScene(sc.flatten()).render(sc.camera, seed=6, window=(40, 30, 80, 60), want_u8=False)
    assert np.array_equal(out["default"][1][30:90, 40:120], rsig)


def test_head_scene_published_size_speed(gpu_ctx):
    """The reference's run() scene at its published 800x800: the whole frame two-stage vs reference walk, bit for bit,
    and the speed-up the two-stage traversal is there for (round 1: 115 Msamples/s, 98 % of it in wf_trav)."""
    sc = scenes.head_scene(800, 800, 64, 10, textures=scenes.load_asset_textures())
    gpu_ctx.upload(sc.flatten())
    f32, _, sig, st = gpu_ctx.render(sc.camera, seed=1, want_u8=False, want_sig=True)
    r32, _, rsig, rst = gpu_ctx.render(sc.camera, seed=1, want_u8=False, want_sig=True, flags=abi.MI_OPT_REFERENCE_WALK)
    assert np.array_equal(sig, rsig) and np.array_equal(f32, r32)
    _, _, _, st = gpu_ctx.render(sc.camera, seed=1, want_u8=False)
    _, _, _, rst = gpu_ctx.render(sc.camera, seed=1, want_u8=False, flags=abi.MI_OPT_REFERENCE_WALK)
    fast, slow = st.samples / st.kernel_ms / 1e3, rst.samples / rst.kernel_ms / 1e3
    print(f"HEAD 800x800x64: two-stage {fast:.0f} Msamples/s ({gpu_ctx.last_pipeline_ms()}), reference walk {slow:.0f} Msamples/s")
    assert fast > 2.0 * slow
