"""Host-side logic above the C ABI (CPU only): OBJ reader (tobj semantics), cgmath helpers,
scene flattening order, tile partition arithmetic and the N>1 gather path on gloo."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from cs397raytracingsp22_amd import (Camera, ConvexVolume, Dielectric, Isotropic, Lambertian, Scene, Sphere, StaticMesh,
                                     Triangle, abi, cgmath, dist, objload, scenes)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_OBJ = "/root/reference/obj"


def test_obj_reader_fan_triangulation_and_single_index():
    text = textwrap.dedent("""
        v 0 0 0
        v 1 0 0
        v 1 1 0
        v 0 1 0
        v 0.5 2 0
        vt 0 0
        vt 1 0
        vt 1 1
        vt 0 1
        vn 0 0 1
        f 1/1/1 2/2/1 3/3/1 4/4/1 5/3/1
        f 1/1/1 3/3/1 -1/4/1
    """)
    m = objload.load_obj_text(text)[0]
    # pentagon -> fan (0,k,k+1): 3 triangles; + 1 triangle
    assert m.n_triangles == 4
    idx = m.indices.reshape(-1, 3)
    assert idx[0].tolist() == [0, 1, 2] and idx[1].tolist() == [0, 2, 3] and idx[2].tolist() == [0, 3, 4]
    # (v5, vt3) and (v5, vt4) are different single-index vertices; negative index = last vertex
    assert m.n_vertices == 6
    assert np.allclose(m.positions.reshape(-1, 3)[idx[3][2]], [0.5, 2, 0])
    assert np.allclose(m.texcoords.reshape(-1, 2)[idx[3][2]], [0, 1])


def test_obj_reader_keeps_first_model_only():
    text = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvn 0 0 1\ng a\nf 1/1/1 2/1/1 3/1/1\ng b\nf 3/1/1 2/1/1 1/1/1\nf 1/1/1 2/1/1 3/1/1\n"
    models = objload.load_obj_text(text)
    assert [m.n_triangles for m in models] == [1, 2]            # the reference uses models.remove(0) (geometry.rs:157)


@pytest.mark.skipif(not os.path.isdir(REF_OBJ), reason="reference assets not present (GPU box)")
@pytest.mark.parametrize("name,tris", [("teapot", 240), ("cube", 12), ("drone", 1736), ("sphere", 32512)])
def test_committed_assets_equal_reader_output(name, tris):
    m = objload.load_obj(os.path.join(REF_OBJ, name + ".obj"))[0]
    a = scenes.load_asset_mesh(name)
    assert m.n_triangles == tris == a.n_triangles
    assert np.array_equal(m.positions, a.positions) and np.array_equal(m.indices, a.indices)
    assert np.array_equal(m.normals, a.normals) and np.array_equal(m.texcoords, a.texcoords)


def test_cgmath_conventions():
    T = cgmath.from_translation((1, 2, 3))
    assert np.allclose(cgmath.cols16(T)[12:15], [1, 2, 3])       # column-major: translation in column 3
    R = cgmath.from_angle_x(90.0)
    assert np.allclose(R @ np.float32([0, 1, 0, 0]), [0, 0, 1, 0], atol=1e-6)     # right-handed, M*v
    Ry = cgmath.from_angle_y(90.0)
    assert np.allclose(Ry @ np.float32([0, 0, 1, 0]), [1, 0, 0, 0], atol=1e-6)
    M = cgmath.mul(T, R, cgmath.from_scale(2.0))
    assert np.allclose(M @ cgmath.inverse_transform(M), np.eye(4), atol=1e-5)


def test_flatten_keeps_scene_object_order_and_shares_materials():
    red = Lambertian(albedo=(1, 0, 0))
    objs = [Sphere((0, 0, 0), 1, red), Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), red),
            ConvexVolume(Sphere((0, 0, 0), 2, Dielectric(1.5)), Isotropic(), 0.5), Sphere((1, 1, 1), 1, Dielectric(1.3))]
    d = Scene(Camera(), objs).flatten().desc
    assert [(d.objects[i].kind, d.objects[i].index) for i in range(d.n_objects)] == [(0, 0), (1, 0), (3, 0), (0, 1)]
    assert d.n_materials == 3 and d.spheres[0].material == d.triangles[0].material
    # a boundary that is not a Sphere travels as a detached entry of its typed array (not listed in Scene.objects)
    d2 = Scene(Camera(), [ConvexVolume(Triangle((0, 0, 0), (1, 0, 0), (0, 1, 0), red), Isotropic(), 1.0)]).flatten().desc
    assert d2.n_objects == 1 and d2.n_triangles == 1 and (d2.volumes[0].boundary_kind, d2.volumes[0].boundary_index) == (abi.MI_OBJ_TRIANGLE, 0)
    with pytest.raises(abi.MiError):          # a medium inside a boundary would draw random numbers inside the boundary query
        Scene(Camera(), [ConvexVolume(ConvexVolume(Sphere((0, 0, 0), 1, red), Isotropic(), 1.0), Isotropic(), 1.0)]).flatten()


def test_tile_partition_arithmetic():
    W, H = 75, 41                                                # ragged: 3 x 2 tiles
    tx, ty, total = dist.tile_grid(W, H)
    assert (tx, ty, total) == (3, 2, 6)
    # the row length of the numbering is coprime with the rank count: 3 columns stay 3 for 2 / 4 / 8 ranks, become 4 for 3 ranks
    assert dist.tile_grid(W, H, 8)[0] == 3 and dist.tile_grid(W, H, 3)[0] == 4 and dist.tile_grid(1920, 1080, 8)[0] == 61
    for world in (1, 2, 3, 4, 5, 8):
        tx, ty, total = dist.tile_grid(W, H, world)
        padded = dist.tiles_padded(W, H, world)
        owned = [dist.tiles_of_rank(W, H, r, world) for r in range(world)]
        assert sorted(sum(owned, [])) == list(range(total))      # a partition
        assert max(len(o) for o in owned) == padded
        rank, idx = dist.compact_index(W, H, world)
        assert idx.max() < padded * dist.TILE_PIXELS
        # (rank, idx) is injective over the image
        key = rank.astype(np.int64) * padded * dist.TILE_PIXELS + idx
        assert np.unique(key).size == W * H


def test_library_partition_equals_the_python_mirror_without_a_gpu():
    """mi_compact_size (mi_rt.cpp tile_counts, what mi_multi_render and the RCCL slice sizes use) against dist.tile_grid for
    widths whose tile-column count is / is not coprime with the world and whose last column is / is not partial.  No GPU."""
    from cs397raytracingsp22_amd import Camera, compact_size
    for W in (32, 75, 192, 203, 224, 250, 256, 1920, 3840):
        for H in (41, 64, 1080):
            cam = Camera(screen_width=W, screen_height=H, aa_sample_count=4)
            for world in (1, 2, 3, 4, 5, 6, 7, 8):
                total, padded = compact_size(cam, world)
                tx, ty, t2 = dist.tile_grid(W, H, world)
                assert total == t2 == tx * ty and padded == dist.tiles_padded(W, H, world), (W, H, world)
                assert np.gcd(tx, world) == 1 and tx >= (W + 31) // 32
                # every rank owns `padded` or `padded - 1` tiles, and never none when there are at least `world` tiles
                counts = [len(dist.tiles_of_rank(W, H, r, world)) for r in range(world)]
                assert max(counts) == padded and min(counts) >= padded - 1


GLOO_WORKER = """
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from cs397raytracingsp22_amd import dist as pdist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
W, H = 75, 41
rng = np.random.default_rng(5)
image = rng.random((H, W, 3)).astype(np.float32)          # the frame every rank would agree on
r_of, idx = pdist.compact_index(W, H, world)
padded = pdist.tiles_padded(W, H, world)
local = np.zeros((padded * pdist.TILE_PIXELS, 3), np.float32)
mine = r_of == rank
local[idx[mine]] = image[mine]                            # what K1 writes for this rank's tiles
g = pdist.gather_compact(torch.from_numpy(local.reshape(padded, pdist.TILE_PIXELS, 3)), world, rank)
if rank == 0:
    flat = g.numpy().reshape(world, -1, 3)
    out = flat[r_of, idx]                                 # K3's mapping (numpy checker)
    assert np.array_equal(out, image), "gathered frame differs"
    print("GLOO_OK", world)
else:
    assert g is None
dist.barrier()
dist.destroy_process_group()
"""


def test_gather_path_world_size_2_gloo(tmp_path):
    """N > 1 host path on CPU: tile ownership, the frame's single gather (gloo here, RCCL on
    the GPU box) and the un-permute mapping reassemble the image exactly."""
    script = tmp_path / "w.py"
    script.write_text(GLOO_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GLOO_OK 2" in out.stdout


def test_png_writer_round_trip(tmp_path):
    from cs397raytracingsp22_amd.image import save_png
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    path = tmp_path / "x.png"
    save_png(str(path), img)
    from PIL import Image
    back = np.asarray(Image.open(path).convert("RGB"))
    assert np.array_equal(back, img)


def test_bench_without_a_launcher_takes_the_in_process_route_and_fails_loudly_without_gpus():
    """`python bench.py --gpus N` (no torch.distributed.run, WORLD_SIZE unset) must go through mi_multi_* in-process — never ask for a
    launcher, never fall back to anything: on a box without devices that is mi_multi_create's own error and a non-zero exit."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present (tests/test_gpu_multi.py covers the route there)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "no CPU fallback" in r.stderr and "torch.distributed.run" not in r.stderr
    assert not any(line.startswith("{") for line in r.stdout.splitlines())      # no result line was printed


def test_bench_line_says_what_the_headline_counts(capsys):
    """bench.report() on fixed numbers (no GPU): the ONE JSON line carries the contract's keys, the dead-tile share, the rate over the
    pixels that see something, the segment rate from the device's own count, and the HBM fraction as a range when a committed PMC
    profile of the same kernel sources exists."""
    import importlib.util
    import json
    import types
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from cs397raytracingsp22_amd import scenes
    sc = scenes.config2()
    flat = sc.flatten()
    args = types.SimpleNamespace(steps=5, config="cfg2", variant=0, spp=0, flags=0, max_state_gb=0.0, no_cpu_baseline=True, warmup=1)
    samples = 1920 * 1080 * 256
    counts = {"passes": 11, "paths_a": 494431863, "paths_b": 338823571, "queue_entries": 338823571, "sample_slots": 534773760,
              "pixels": 2088960, "segments": 1674413948, "dead_tile_samples": 212336640}
    pipe = {"wf_main_ms": 260.0, "wf_trav_ms": 120.0, "wf_reduce_ms": 4.6, "launches": 180, "wf_trav_f_ms": 0.0, "wf_replay_ms": 0.0, "wf_main_a_ms": 240.0}
    bench.report(args, sc, flat, 1, None, "one process, one GPU (mi_ctx_*)", 0.385, 76.8, pipe, counts)
    rec = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in rec, key
    assert rec["unit"] == "Msamples/s" and rec["dtype"] == "f32" and rec["vs_baseline"] is None and rec["n_gpus"] == 1
    assert abs(rec["value"] - samples * 5 / 0.385 / 1e6) < 1e-6 * rec["value"]
    cfg = rec["config"]
    assert abs(cfg["dead_tile_frac"] - 212336640 / samples) < 1e-12
    assert abs(cfg["live_pixel_msamples_per_s"] - (samples - 212336640) * 5 / 0.385 / 1e6) < 1e-6 * cfg["live_pixel_msamples_per_s"]
    assert abs(cfg["segments_per_sample"] - 1674413948 / samples) < 1e-12 and cfg["msegments_per_s"] > rec["value"]
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["unit"] == "GB/s" and 0 < roof["frac"] <= 1
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["traffic"] > 0
    if roof["frac_raw_counters"] is not None:           # a committed PMC profile of these kernel sources: the fraction is a range
        assert roof["frac_raw_counters"] < roof["frac_x2_corrected"] == roof["frac"] and "committed rocprofv3 PMC" in roof["traffic_source"]
    else:
        assert "traffic model" in roof["traffic_source"]
    # N > 1 ranks: the per-device counts are device 0's only, so the whole-frame figures derived from them are left out
    bench.report(args, sc, flat, 8, "nccl", "one process per GPU", 0.385, 76.8, pipe, counts)
    rec8 = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert rec8["n_gpus"] == 8 and rec8["config"]["dead_tile_frac"] is None and rec8["config"]["msegments_per_s"] is None and "cpu_baseline" not in rec8
