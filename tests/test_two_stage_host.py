"""The exact two-stage mesh traversal, host side (CPU only): the F-tree builder and the padding bound of
cs397raytracingsp22_amd/csrc/bvh_build.hpp — the code the scene compiler runs — checked by tests/cpp/two_stage_check.cpp:
pass 1 (padded F-tree walk + the reference's own triangle test) followed by pass 2 (the reference's walk replayed over
the root-to-candidate paths) must return the reference walk's (distance bits, triangle) for every ray, including rays
lying almost in a triangle's plane, rays through vertices and along edges, axis-aligned rays and tiny / huge |d|."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSETS = os.path.join(ROOT, "cs397raytracingsp22_amd", "assets")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = tmp_path_factory.mktemp("ts") / "two_stage_check"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall",
                    os.path.join(ROOT, "tests", "cpp", "two_stage_check.cpp"), "-o", str(exe)], check=True)
    return str(exe)


def mesh_bin(tmp_path, name):
    z = np.load(os.path.join(ASSETS, name + ".npz"), allow_pickle=False)
    pos = z["positions"].astype(np.float32).ravel()
    idx = z["indices"].astype(np.uint32).ravel()
    p = tmp_path / (name + ".bin")
    with open(p, "wb") as fh:
        fh.write(struct.pack("ii", len(pos) // 3, len(idx) // 3))
        fh.write(pos.tobytes())
        fh.write(idx.tobytes())
    return str(p)


@pytest.mark.parametrize("name,scale,rays,offset", [("teapot", 1.0, 200000, 0.0), ("cube", 1.0, 100000, 0.0), ("sphere", 1.0, 150000, 0.0),
                                                   ("sphere", 0.01, 60000, 0.0), ("sphere", 40.0, 60000, 0.0), ("teapot", 0.05, 60000, 0.0),
                                                   ("drone", 1.0, 60000, 0.0), ("drone", 0.001, 100000, 0.0),
                                                   # far from the object-space origin: |o| >> padding (the padding must survive f32 rounding)
                                                   ("sphere", 1.0, 100000, 3000.0), ("teapot", 0.2, 100000, -700.0), ("sphere", 0.02, 60000, 50.0),
                                                   ("sphere", 20.0, 60000, 300000.0)])
def test_two_stage_equals_the_reference_walk(checker, tmp_path, name, scale, rays, offset):
    r = subprocess.run([checker, mesh_bin(tmp_path, name), str(scale), str(rays), "7", str(offset)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:] + r.stdout[-500:]
    st = json.loads(r.stdout)
    assert st["mismatches"] == 0 and st["entered"] > rays // 10
    if name == "drone" and scale == 1.0:
        # object-space scale makes the 1e-4 determinant test void: the bound must refuse (nearly) every ray: reference walk
        assert st["fallback"] >= 0.99 * st["entered"]          # (a few rays with a tiny |d| are covered)
    if name == "sphere" and scale == 1.0 and offset == 0.0:
        # the point of the exercise: obj/sphere.obj's file order defeats the reference's index-range tree
        assert st["fallback"] == 0
        assert st["ref_box_per_entry"] > 1000 and st["f_nodes_per_entry"] + 3 * st["f_tri_per_entry"] + st["replay_slabs_per_entry"] < 300
    print(name, scale, st)
