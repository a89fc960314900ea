"""Progressive / resumable accumulation (mi_render_samples_device): slicing the
reference's per-pixel sample loop (tracing.rs:233-241) must not change its result."""
import ctypes as C

import numpy as np
import pytest

from cs397raytracingsp22_amd import Context, abi, scenes
from cs397raytracingsp22_amd.progressive import ProgressiveRender

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["config2", "config5", "head"])
def test_slices_with_checkpoint_equal_one_call(gpu_ctx, tmp_path, name):
    sc = {"config2": lambda: scenes.config2(160, 96, 16, 10), "config5": lambda: scenes.config5(96, 64, 25, 50),
          "head": lambda: scenes.head_scene(80, 64, 12, 10)}[name]()         # head: meshes of both kinds (reference walk + two-stage)
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    want, _, want_sig, _ = gpu_ctx.render(sc.camera, seed=9, want_u8=False, want_sig=True)

    pr = ProgressiveRender(gpu_ctx, sc.camera, seed=9, want_sig=True)
    st = pr.advance(3)
    assert st.samples == 3 * sc.camera.screen_width * sc.camera.screen_height and pr.done == 3 and not pr.finished
    prev = pr.preview()
    assert prev.shape == want.shape and np.isfinite(prev).all() and prev.max() > 0
    pr.advance(6)
    ckpt = str(tmp_path / "half.npz")
    pr.save(ckpt)

    ctx2 = Context(0)                                   # "another process": fresh context, state from disk
    try:
        ctx2.upload(flat)
        pr2 = ProgressiveRender.resume(ctx2, sc.camera, ckpt, want_sig=True)
        assert pr2.done == 9
        pr2.advance(10 ** 6)                            # clamped to what is left
        assert pr2.finished and pr2.advance(1) is None
        got, got_sig = pr2.result()
    finally:
        ctx2.close()
    assert np.array_equal(got_sig, want_sig)
    assert np.array_equal(got, want)                    # same additions in the same order: bit-identical

    with pytest.raises(ValueError):
        other = scenes.config2(160, 96, 16, 9).camera   # different path_depth
        ProgressiveRender.resume(gpu_ctx, other, ckpt)


def test_sample_range_errors(gpu_ctx):
    import torch
    sc = scenes.config1(64, 64, 4, 4)
    gpu_ctx.upload(sc.flatten())
    acc = torch.zeros((4 * 1024, 4), dtype=torch.float32, device="cuda:0")
    out = torch.zeros((4 * 1024, 3), dtype=torch.float32, device="cuda:0")
    for b, e, code in ((2, 2, abi.MI_ERR_INVALID), (3, 2, abi.MI_ERR_INVALID), (0, 5, abi.MI_ERR_INVALID)):
        with pytest.raises(abi.MiError) as ei:
            gpu_ctx.render_samples_device(sc.camera, b, e, acc.data_ptr(), out.data_ptr())
        assert ei.value.code == code
    with pytest.raises(abi.MiError) as ei:              # the final slice needs the output buffer
        gpu_ctx.render_samples_device(sc.camera, 0, 4, acc.data_ptr(), None)
    assert ei.value.code == abi.MI_ERR_INVALID
    with pytest.raises(abi.MiError) as ei:
        gpu_ctx.render_samples_device(sc.camera, 0, 2, None, None)
    assert ei.value.code == abi.MI_ERR_INVALID
    gpu_ctx.render_samples_device(sc.camera, 0, 2, acc.data_ptr(), None)      # partial slice: no output needed
    gpu_ctx.render_samples_device(sc.camera, 2, 4, acc.data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    assert float(out.abs().sum()) > 0
