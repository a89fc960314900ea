"""Multi-GPU behind the C ABI (mi_multi_*: one context + stream + host thread per device inside the library, tiles
t % N, ONE RCCL send/recv fan-in per frame, K3 + K4 on device 0).  On the one-GPU box: a communicator of one device
(ncclCommInitAll runs, the exchange is empty) must give mi_render's image bit for bit.  The N >= 2 cases run on the
first lease that has the devices; they are skipped, not faked, elsewhere."""
import numpy as np
import pytest

from cs397raytracingsp22_amd import Context, MultiContext, scenes

pytestmark = pytest.mark.gpu


def _devices():
    import torch
    return torch.cuda.device_count()


def test_multi_one_device_equals_mi_render(gpu_ctx):
    sc = scenes.config2(203, 117, 16, 10)            # ragged: partial edge tiles
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    ref32, ref8, refsig, rst = gpu_ctx.render(sc.camera, seed=3, want_sig=True)
    m = MultiContext(1)
    try:
        m.upload(flat)
        m.reserve(sc.camera)
        f32, u8, sig, st = m.render(sc.camera, seed=3, want_sig=True)
        again32, _, _, _ = m.render(sc.camera, seed=3, want_u8=False)
    finally:
        m.close()
    assert np.array_equal(f32, ref32) and np.array_equal(u8, ref8) and np.array_equal(sig, refsig)
    assert np.array_equal(again32, ref32)
    assert st.samples == rst.samples and st.kernel_ms > 0 and st.total_ms >= st.kernel_ms * 0.5


def test_multi_one_device_with_meshes_of_both_kinds(gpu_ctx):
    """The reference's run() scene (drone + cube through the reference walk, the sphere two-stage: wf_filter_f, wf_trav_f on its own
    stream, wf_replay) through mi_multi_* on its per-device stream: the same image as mi_render."""
    sc = scenes.head_scene(96, 80, 8, 10)
    flat = sc.flatten()
    gpu_ctx.upload(flat)
    ref32, _, refsig, _ = gpu_ctx.render(sc.camera, seed=8, want_u8=False, want_sig=True)
    m = MultiContext(1)
    try:
        m.upload(flat)
        f32, _, sig, _ = m.render(sc.camera, seed=8, want_u8=False, want_sig=True)
    finally:
        m.close()
    assert np.array_equal(sig, refsig) and np.array_equal(f32, ref32, equal_nan=True)


def test_multi_error_paths():
    from cs397raytracingsp22_amd import abi
    with pytest.raises(abi.MiError) as ei:
        MultiContext(0)
    assert ei.value.code == abi.MI_ERR_INVALID
    with pytest.raises(abi.MiError):
        MultiContext(2, devices=[0, 0])               # the same device twice
    m = MultiContext(1)
    try:
        with pytest.raises(abi.MiError) as ei:
            m.render(scenes.config1(64, 64, 4, 4).camera)
        assert ei.value.code == abi.MI_ERR_NO_SCENE
    finally:
        m.close()


@pytest.mark.parametrize("n", [2, 4, 8])
def test_multi_n_devices_equals_one_device(n):
    """RCCL over xGMI for real: N devices in one process, image identical to the one-device render."""
    if _devices() < n:
        pytest.skip(f"needs {n} GPUs, this box has {_devices()}")
    sc = scenes.config2(640, 360, 16, 10)
    flat = sc.flatten()
    one = Context(0)
    try:
        one.upload(flat)
        ref32, ref8, refsig, _ = one.render(sc.camera, seed=5, want_sig=True)
    finally:
        one.close()
    m = MultiContext(n)
    try:
        m.upload(flat)
        f32, u8, sig, st = m.render(sc.camera, seed=5, want_sig=True)
    finally:
        m.close()
    assert np.array_equal(f32, ref32) and np.array_equal(u8, ref8) and np.array_equal(sig, refsig)


def test_torch_nccl_gather_world2():
    """dist.py's path (one process per GPU, torch.distributed backend nccl = RCCL): bench.py --gpus 2 on two real devices."""
    if _devices() < 2:
        pytest.skip("needs 2 GPUs")
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "16"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    rec = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["backend"] == "nccl" and rec["value"] > 0


def _run_bench(*extra, timeout=900):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}      # no launcher
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *extra], capture_output=True, text=True,
                         timeout=timeout, cwd=root, env=env)
    rec = None
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            rec = json.loads(ln)
    return out, rec


def test_bench_in_process_route_one_device():
    """`python bench.py --gpus 1 --via-multi`: the route every N > 1 run without a launcher takes (mi_multi_* in this process,
    no torch), on the one device this box has.  (Bit-identity of that route with mi_render: the first test of this file.)"""
    out, rec = _run_bench("--gpus", "1", "--via-multi", "--steps", "2", "--warmup", "1", "--spp", "16", "--no-cpu-baseline")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert rec["n_gpus"] == 1 and rec["backend"].startswith("rccl (mi_multi") and rec["config"]["ranks"] == 1
    assert rec["value"] > 0 and rec["roofline"]["frac"] is not None and 0 < rec["roofline"]["frac"] <= 1
    assert rec["roofline"]["path_counts"]["passes"] > 0 and rec["steps"] == 2 and rec["warmup"] == 1


def test_bench_more_gpus_than_devices_fails_loudly():
    """`python bench.py --gpus N` with N > devices: mi_multi_create's own error and a non-zero exit — not a hint to use a launcher,
    not a silent N = 1 run."""
    n = _devices() + 1
    out, rec = _run_bench("--gpus", str(n), "--steps", "1", "--warmup", "0", "--spp", "16", "--no-cpu-baseline", timeout=300)
    assert out.returncode != 0 and rec is None
    assert "out of range" in out.stderr and "torch.distributed.run" not in out.stderr


@pytest.mark.parametrize("n", [2, 4, 8])
def test_bench_in_process_n_devices(n):
    """The scaling bench exactly as `python bench.py --gpus N` runs it with no launcher: native RCCL fan-in inside mi_multi_render."""
    if _devices() < n:
        pytest.skip(f"needs {n} GPUs, this box has {_devices()}")
    out, rec = _run_bench("--gpus", str(n), "--steps", "2", "--warmup", "1", "--spp", "64", "--no-cpu-baseline")
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert rec["n_gpus"] == n and rec["config"]["ranks"] == n and rec["backend"].startswith("rccl (mi_multi") and rec["value"] > 0
