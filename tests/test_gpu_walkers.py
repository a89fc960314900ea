"""Every storage mode of the reference-tree walker gives the same paths: BVH in global memory (0), whole image in LDS with 256-thread
blocks (2) or one 1024-thread block per CU (3), and wf_trav_i — interior nodes in LDS, leaves from global memory, two 1024-thread
blocks per CU (4: what trees of 64 .. 150 KB such as obj/drone.obj take by default), with the leaf records in LDS too (5: trees whose whole
split image fits 64 KB), or that with the interior records in the paired {near, far} layout (6: the default for SEVERAL small trees; one small tree - the
teapot - takes mode 5, whose single-mesh form tests boxes with the clamped v_med3 form).  The mode is a developer knob read once in
mi_ctx_create (MI_RT_WF_TRAV_LDS), so each mode gets a context of its own; signatures must equal the oracle's bit for bit."""
import os

import numpy as np
import pytest

from cs397raytracingsp22_amd import Context, Lambertian, StaticMesh, abi, cgmath, scenes

pytestmark = pytest.mark.gpu


def render_with_env(env, flat, cam, seed, flags=0):
    """A context of its own created under the given developer knobs (read once in mi_ctx_create)."""
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        ctx = Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    try:
        ctx.upload(flat)
        f32, _, sig, _ = ctx.render(cam, seed=seed, want_sig=True, flags=flags)
        return f32, sig
    finally:
        ctx.close()


def render_with_mode(mode, flat, cam, seed):
    return render_with_env({"MI_RT_WF_TRAV_LDS": mode}, flat, cam, seed, flags=abi.MI_OPT_REFERENCE_WALK)


def test_walker_storage_modes_agree_with_the_oracle_teapot(orc):
    sc = scenes.config2(144, 96, 16, 10)
    flat = sc.flatten()
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=4)
    ref = None
    for mode in (0, 2, 3, 4, 5, 6):
        f32, sig = render_with_mode(mode, flat, sc.camera, 4)
        assert int((sig != rsig).sum()) == 0, f"walker mode {mode}: paths differ from the oracle"
        ref = f32 if ref is None else ref
        assert np.array_equal(f32, ref), mode


def test_walker_storage_modes_agree_several_meshes_and_a_leaf_root(orc):
    """Teapot + cube + a one-triangle mesh (its root is a leaf: never box-tested, geometry.rs:95) + the teapot again."""
    from test_oracle_kat import cube_mesh
    from test_gpu_edge_cases import tiny_mesh
    sc = scenes.config2(128, 96, 16, 8)
    grey = Lambertian(albedo=(0.7, 0.7, 0.7))
    sc.objects.append(StaticMesh(cube_mesh(-0.5, 0.5), grey, [None] * 5, cgmath.from_translation((-1.8, 0.5, 1.0))))
    sc.objects.append(StaticMesh(tiny_mesh(1), grey, [None] * 5, cgmath.from_translation((1.5, 2.5, 0.5))))
    sc.objects.append(sc.objects[-3])                                     # the teapot once more (shared StaticMesh)
    flat = sc.flatten()
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=11)
    for mode in (2, 4, 5, 6):
        _, sig = render_with_mode(mode, flat, sc.camera, 11)
        assert int((sig != rsig).sum()) == 0, f"walker mode {mode}: paths differ from the oracle"


def test_drone_takes_the_interior_in_lds_walker_and_matches(orc, gpu_ctx):
    """cfg4's tree (3471 nodes: 140 KB as a whole image, 55 KB of interior nodes) at a small size: default mode against mode 3."""
    sc = scenes.config4(96, 64, 16, 6, tex_size=64)
    flat = sc.flatten()
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=2)
    gpu_ctx.upload(flat)
    _, _, sig, _ = gpu_ctx.render(sc.camera, seed=2, want_sig=True)
    assert int((sig != rsig).sum()) == 0
    _, sig3 = render_with_mode(3, flat, sc.camera, 2)
    assert np.array_equal(sig3, rsig)


@pytest.mark.parametrize("scene", ["cfg1", "cfg2", "cfg5", "head"])
def test_pass_schedules_agree_with_the_oracle(orc, scene):
    """The pass-by-pass schedule (every pass through the HBM path state, one or two in-launch rounds), the tail schedule (a pass
    that no longer fills the chip runs every path as far as it can inside one launch — what frames this small take by default)
    and a schedule that waits for every header give the same paths."""
    sc = {"cfg1": lambda: scenes.config1(160, 96, 16, 8), "cfg2": lambda: scenes.config2(160, 96, 16, 10),
          "cfg5": lambda: scenes.config5(96, 64, 8, 50), "head": lambda: scenes.head_scene(96, 96, 8, 10)}[scene]()
    flat = sc.flatten()
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=9)
    ref = None
    for env in ({"MI_RT_WF_TAIL_PATHS": 0}, {"MI_RT_WF_TAIL_PATHS": 4000}, {"MI_RT_WF_TAIL_PATHS": 1 << 22},
                {"MI_RT_WF_TAIL_PATHS": 0, "MI_RT_WF_NOWAIT_BLOCKS": 0}):
        f32, sig = render_with_env(env, flat, sc.camera, 9)
        assert int((sig != rsig).sum()) == 0, f"{env}: paths differ from the oracle"
        ref = f32 if ref is None else ref
        assert np.array_equal(f32, ref), env


@pytest.mark.parametrize("mode", [0, 2, 3, 4, 5, 6])
def test_axis_aligned_rays_through_meshes(orc, mode):
    """Rays with exactly zero direction components: 1/d is infinite on two axes and the slab test meets NaN products (0 * inf) at
    box planes through the ray, which the reference's f32::max / min drop (geometry.rs:59-76).  An orthographic camera along -z
    sends every camera ray with d = (0, 0, -1) into meshes that are only translated, so object space sees the same direction; a
    mirror cube reflects some of them straight back.  (Written for an experimental v_med3 form of the walkers' box test, which
    needs another form for exactly these rays — tools/experiments/r03_walker_med3_slab.diff; kept as a parity case of its own.)"""
    from test_oracle_kat import cube_mesh
    from cs397raytracingsp22_amd import Metal
    sc = scenes.config2(96, 64, 8, 6)
    sc.camera.projection_mode = abi.MI_PROJ_ORTHOGRAPHIC
    sc.camera.view_dir = (0.0, 0.0, -1.0)
    grey = Lambertian(albedo=(0.7, 0.7, 0.7))
    sc.objects.append(StaticMesh(cube_mesh(-0.5, 0.5), grey, [None] * 5, cgmath.from_translation((0.0, 2.0, 0.5))))
    sc.objects.append(StaticMesh(cube_mesh(-0.25, 0.25), Metal(albedo=(0.9, 0.9, 0.9), roughness=0.0), [None] * 5, cgmath.from_translation((0.5, 2.0, 1.5))))
    flat = sc.flatten()
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=21)
    _, sig = render_with_mode(mode, flat, sc.camera, 21)
    assert int((sig != rsig).sum()) == 0, f"walker mode {mode}: paths differ from the oracle"


@pytest.mark.parametrize("mode", [4, 5, 6])
def test_axis_aligned_rays_through_one_mesh(orc, mode):
    """The same rays against ONE mesh: the single-mesh forms of wf_trav_i over 32-byte records (modes 4 and 5) test boxes with the clamped
    form (slab_med3) and must send a wave that holds a ray with a zero direction component through the reference form instead — here every
    camera ray is such a ray, and the mirror cube returns some of them along the axis.  Mode 6 (paired records) has no such split."""
    from test_oracle_kat import cube_mesh
    from cs397raytracingsp22_amd import Metal
    sc = scenes.config1(96, 64, 8, 6)                                      # the Cornell box without the teapot
    sc.camera.projection_mode = abi.MI_PROJ_ORTHOGRAPHIC
    sc.camera.view_dir = (0.0, 0.0, -1.0)
    sc.objects.append(StaticMesh(cube_mesh(-0.6, 0.6), Metal(albedo=(0.9, 0.9, 0.9), roughness=0.0), [None] * 5, cgmath.from_translation((0.25, 2.0, 0.5))))
    flat = sc.flatten()
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=23)
    _, sig = render_with_mode(mode, flat, sc.camera, 23)
    assert int((sig != rsig).sum()) == 0, f"walker mode {mode}: paths differ from the oracle"


def test_clamped_box_test_on_the_drone_with_grazing_and_axis_parallel_rays(orc):
    """cfg4's tree through its default walker (mode 4, one mesh: the clamped box test) under an orthographic camera along -x: every camera ray
    has two zero direction components (the reference form, rolled loop), the bounce rays have none (the clamped form) - both inside one render."""
    sc = scenes.config4(80, 56, 8, 6, tex_size=32)
    sc.camera.projection_mode = abi.MI_PROJ_ORTHOGRAPHIC
    sc.camera.view_dir = (-1.0, 0.0, 0.0)
    sc.camera.up = (0.0, 1.0, 0.0)
    flat = sc.flatten()
    _, _, rsig, _ = orc.OracleScene(flat).render(sc.camera, seed=5)
    f32, sig = render_with_env({}, flat, sc.camera, 5)
    assert int((sig != rsig).sum()) == 0
