"""mi_multi_render's N >= 2 code on the one GPU there is: the loopback test transport (mi_multi_create_loopback).

What a world of N ranks adds to a render — one host thread per rank driving its own wavefront pipeline (each polling its own
pinned header ring) on its own three streams, the coprime tile grid, the `r * slice` receive offsets of the fan-in, the
un-permute and the tone-map with world > 1, the max-over-ranks statistics — runs here exactly as it does over RCCL; only the
transfer itself is a device-to-device copy ordered by an event instead of an ncclSend / ncclRecv pair.  The replacement for
rayon's row split (tracing.rs:228) must not change a bit of the image: every case compares with mi_render on one context.

Also here: two plain mi_ctx driven from two host threads at once (include/mi_rt.h: "distinct contexts may be used from
distinct threads")."""
import threading

import numpy as np
import pytest

from cs397raytracingsp22_amd import Context, MultiContext, abi, scenes

pytestmark = pytest.mark.gpu


def _reference(sc, flat, seed, **kw):
    one = Context(0)
    try:
        one.upload(flat)
        return one.render(sc.camera, seed=seed, want_sig=True, **kw)
    finally:
        one.close()


@pytest.fixture(scope="module")
def cfg2_ragged():
    sc = scenes.config2(203, 117, 16, 10)            # 7 x 4 tiles, partial right and bottom edges
    flat = sc.flatten()
    return sc, flat, _reference(sc, flat, 3)


@pytest.fixture(scope="module")
def head():
    sc = scenes.head_scene(96, 80, 8, 10)             # run()'s scene: reference walk + two-stage meshes, volumes, a plane
    flat = sc.flatten()
    return sc, flat, _reference(sc, flat, 8)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_loopback_cfg2_ragged(cfg2_ragged, n):
    sc, flat, (ref32, ref8, refsig, rst) = cfg2_ragged
    m = MultiContext.loopback(n)
    try:
        assert m._lib.mi_multi_device_count(m._h) == n
        m.upload(flat)
        m.reserve(sc.camera)
        f32, u8, sig, st = m.render(sc.camera, seed=3, want_sig=True)
        again32, again8, _, _ = m.render(sc.camera, seed=3)          # second frame through the same buffers, no signatures
    finally:
        m.close()
    assert np.array_equal(sig, refsig)
    assert np.array_equal(f32, ref32) and np.array_equal(u8, ref8)
    assert np.array_equal(again32, ref32) and np.array_equal(again8, ref8)
    # every sample of the image was traced exactly once (the surplus columns of the coprime grid hold no pixel)
    assert st.samples == rst.samples == 203 * 117 * 16
    total, padded = abi_compact(sc.camera, n)
    assert st.tiles_padded == padded and st.tiles <= total and st.kernel_ms > 0


def abi_compact(cam, world):
    from cs397raytracingsp22_amd import compact_size
    return compact_size(cam, world)


@pytest.mark.parametrize("n", [2, 3, 8])
def test_loopback_head_scene_both_walker_kinds(head, n):
    sc, flat, (ref32, ref8, refsig, _) = head
    m = MultiContext.loopback(n)
    try:
        m.upload(flat)
        f32, u8, sig, _ = m.render(sc.camera, seed=8, want_sig=True)
    finally:
        m.close()
    assert np.array_equal(sig, refsig)
    assert np.array_equal(f32, ref32, equal_nan=True) and np.array_equal(u8, ref8)


def test_loopback_multibatch_and_forced_two_stage():
    """Several sample batches per rank (a small budget) and the two-stage walk forced for every mesh: the ranks' pipelines go
    through wf_reduce's first / middle / last forms side by side."""
    sc = scenes.config2(320, 200, 64, 10)
    flat = sc.flatten()
    ref32, ref8, refsig, _ = _reference(sc, flat, 11)
    m = MultiContext.loopback(3)
    try:
        m.upload(flat)
        f32, u8, sig, st = m.render(sc.camera, seed=11, want_sig=True, max_state_bytes=96 << 20)
        g32, _, gsig, _ = m.render(sc.camera, seed=11, want_sig=True, want_u8=False, flags=abi.MI_OPT_TWO_STAGE)
    finally:
        m.close()
    assert np.array_equal(sig, refsig) and np.array_equal(f32, ref32) and np.array_equal(u8, ref8)
    assert np.array_equal(gsig, refsig) and np.array_equal(g32, ref32)


def test_loopback_resident_outputs_and_per_rank_queries(cfg2_ragged):
    """No output pointer: the frame stays on device 0 (nothing over PCIe) and the call still works; the per-rank contexts
    answer the pipeline queries afterwards."""
    sc, flat, (ref32, _, _, rst) = cfg2_ragged
    m = MultiContext.loopback(4)
    try:
        m.upload(flat)
        _, _, _, st = m.render(sc.camera, seed=3, want_f32=False, want_u8=False)
        paths = 0
        for r in range(4):
            c = m.context(r)
            assert c.last_kernel_ms() > 0
            paths += c.last_pipeline_counts()["sample_slots"]
        f32, _, _, _ = m.render(sc.camera, seed=3, want_u8=False)
    finally:
        m.close()
    assert st.samples == rst.samples
    # sample slots over the ranks: every tile slot is MI_TILE^2 pixels x spp, including the padding
    assert paths >= rst.samples
    assert np.array_equal(f32, ref32)


def test_loopback_error_paths():
    with pytest.raises(abi.MiError) as ei:
        MultiContext.loopback(0)
    assert ei.value.code == abi.MI_ERR_INVALID
    with pytest.raises(abi.MiError) as ei:
        MultiContext.loopback(2, device=99)
    assert ei.value.code == abi.MI_ERR_INVALID
    m = MultiContext.loopback(2)
    try:
        with pytest.raises(abi.MiError) as ei:
            m.render(scenes.config1(64, 64, 4, 4).camera)
        assert ei.value.code == abi.MI_ERR_NO_SCENE
    finally:
        m.close()


def test_two_contexts_from_two_threads(cfg2_ragged, head):
    """Two independent mi_ctx, each used by its own host thread, rendering different scenes at the same time (ctypes releases
    the GIL for the duration of mi_render): both images equal their single-threaded renders, several frames in a row."""
    jobs = [(cfg2_ragged, 3), (head, 8)]
    results = [None, None]
    errors = []
    start = threading.Barrier(2)

    def work(i):
        (sc, flat, _), seed = jobs[i]
        try:
            ctx = Context(0)
            try:
                ctx.upload(flat)
                start.wait(timeout=120)
                frames = [ctx.render(sc.camera, seed=seed, want_sig=True) for _ in range(3)]
            finally:
                ctx.close()
            results[i] = frames
        except Exception as e:          # noqa: BLE001 - reported in the main thread
            errors.append((i, repr(e)))
            try:
                start.abort()
            except Exception:           # noqa: BLE001
                pass

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    for i, ((_, _, (ref32, ref8, refsig, _)), _) in enumerate(jobs):
        assert results[i] is not None
        for f32, u8, sig, _ in results[i]:
            assert np.array_equal(sig, refsig)
            assert np.array_equal(f32, ref32, equal_nan=True) and np.array_equal(u8, ref8)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_in_process_route_with_loopback_ranks(n):
    """`python bench.py --gpus N --loopback`: the in-process route of the scaling bench (mi_multi_* behind the ABI, what `python bench.py
    --gpus N` runs with no launcher) with N ranks on this box's one device.  The line says what it is: not a scaling figure."""
    from test_gpu_multi import _run_bench
    out, rec = _run_bench("--gpus", str(n), "--loopback", "--steps", "2", "--warmup", "1", "--spp", "16", "--no-cpu-baseline")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert rec["n_gpus"] == n and rec["config"]["ranks"] == n and rec["backend"].startswith("loopback test transport")
    assert rec["value"] > 0 and rec["steps"] == 2 and rec["config"]["mi_multi_total_ms"] > 0


@pytest.mark.parametrize("case", ["voted", "recursive", "phong", "orthographic"])
def test_loopback_other_kernels_and_camera_modes(case):
    """The kernels that are not the wavefront pipeline go through mi_multi_render's ranks as well: the single-launch megakernel
    (queued, never synchronising by itself), the recursive estimator (path_samples = 2), the Phong debug shader, and an
    orthographic camera (tile masks off).  Three ranks on the one device, bit-identical to mi_render."""
    sc = scenes.config2(150, 100, 4, 6)
    kw = {}
    if case == "voted":
        kw["variant"] = abi.MI_VARIANT_VOTED
    elif case == "recursive":
        sc.camera.path_samples = 2
        sc.camera.path_depth = 4
    elif case == "phong":
        sc.camera.shading_mode = abi.MI_SHADE_PHONG
    else:
        sc.camera.projection_mode = abi.MI_PROJ_ORTHOGRAPHIC
    flat = sc.flatten()
    one = Context(0)
    try:
        one.upload(flat)
        ref32, ref8, refsig, _ = one.render(sc.camera, seed=13, want_sig=True, **kw)
    finally:
        one.close()
    m = MultiContext.loopback(3)
    try:
        m.upload(flat)
        f32, u8, sig, _ = m.render(sc.camera, seed=13, want_sig=True, **kw)
    finally:
        m.close()
    assert np.array_equal(sig, refsig) and np.array_equal(f32, ref32, equal_nan=True) and np.array_equal(u8, ref8)
