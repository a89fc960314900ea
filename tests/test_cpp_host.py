"""The C++ host mirror of the reference interface (cs397raytracingsp22_amd/host/*.hpp) and its
driver mi_rt_cli: compiled code above the C ABI, as the reference is compiled code."""
import os
import subprocess

import numpy as np
import pytest

from cs397raytracingsp22_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "cs397raytracingsp22_amd", "lib", "mi_rt_cli")


def read_ppm(path):
    with open(path, "rb") as fh:
        assert fh.readline().strip() == b"P6"
        w, h = map(int, fh.readline().split())
        assert fh.readline().strip() == b"255"
        return np.frombuffer(fh.read(), np.uint8).reshape(h, w, 3)


def write_obj(mesh, path):
    p, n, t, i = mesh.positions.reshape(-1, 3), mesh.normals.reshape(-1, 3), mesh.texcoords.reshape(-1, 2), mesh.indices.reshape(-1, 3)
    with open(path, "w") as fh:
        for v in p:
            fh.write("v %r %r %r\n" % tuple(float(x) for x in v))
        for v in t:
            fh.write("vt %r %r\n" % tuple(float(x) for x in v))
        for v in n:
            fh.write("vn %r %r %r\n" % tuple(float(x) for x in v))
        for a, b, c in i + 1:
            fh.write(f"f {a}/{a}/{a} {b}/{b}/{b} {c}/{c}/{c}\n")


def test_cli_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([CLI, str(tmp_path / "x.ppm"), "32", "32", "4", "4"], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr
    assert not (tmp_path / "x.ppm").exists()


@pytest.mark.gpu
def test_cpp_caller_matches_python_caller(gpu_ctx, tmp_path):
    out = tmp_path / "c1.ppm"
    r = subprocess.run([CLI, str(out), "96", "80", "16", "8"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "caller=c++" in r.stdout and "gpus=1" in r.stdout
    sc = scenes.config1(96, 80, 16, 8)
    gpu_ctx.upload(sc.flatten())
    _, u8, _, _ = gpu_ctx.render(sc.camera, seed=1)
    assert np.array_equal(read_ppm(out), u8)               # same PODs -> same bytes


@pytest.mark.gpu
def test_cpp_caller_with_obj_mesh(gpu_ctx, tmp_path):
    obj = tmp_path / "teapot.obj"
    write_obj(scenes.load_asset_mesh("teapot"), obj)
    out = tmp_path / "c2.ppm"
    r = subprocess.run([CLI, str(out), "96", "64", "16", "10", str(obj)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sc = scenes.config2(96, 64, 16, 10)
    gpu_ctx.upload(sc.flatten())
    _, u8, _, _ = gpu_ctx.render(sc.camera, seed=1)
    # both mirrors compose the transform and its inverse with the same f32 / f64 operations (cgmath.py mul, inverse_transform
    # = host/geometry.hpp Matrix4): the library receives the same floats, the image is the same bytes
    assert np.array_equal(read_ppm(out), u8)


@pytest.mark.gpu
def test_cpp_caller_decodes_texture_files_like_the_python_mirror(gpu_ctx, tmp_path):
    """StaticMesh::load_from_file with an albedo PNG and a normal-map JPEG (the reference's cube, tracing.rs:385-394): the compiled
    mirror decodes the files itself (host/texture.hpp), the Python mirror through PIL — same texels, same transform, same image.
    The files are written here (the reference's own texture/ directory does not exist on the GPU box)."""
    from PIL import Image
    from cs397raytracingsp22_amd import StaticMesh, cgmath
    from test_oracle_kat import cube_mesh
    rng = np.random.default_rng(12)
    yy, xx = np.mgrid[0:96, 0:128]
    alb = np.stack([(xx * 2) % 256, (yy * 3 + 40) % 256, ((xx + yy) * 5) % 256], axis=2).astype(np.uint8)
    alb[::5, ::7] = rng.integers(0, 256, alb[::5, ::7].shape, dtype=np.uint8)
    nrm = np.stack([128 + 60 * np.sin(xx / 9.0), 128 + 60 * np.cos(yy / 7.0), np.full(xx.shape, 230.0)], axis=2).astype(np.uint8)
    a_png, n_jpg, obj = tmp_path / "albedo.png", tmp_path / "normal.jpg", tmp_path / "cube.obj"
    Image.fromarray(alb, "RGB").save(a_png)
    Image.fromarray(nrm, "RGB").save(n_jpg, quality=92, subsampling=2)
    write_obj(cube_mesh(), obj)
    out = tmp_path / "c3.ppm"
    r = subprocess.run([CLI, str(out), "128", "96", "16", "6", str(obj), "0", str(a_png), str(n_jpg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    sc = scenes.config1(128, 96, 16, 6)
    sc.objects.append(StaticMesh.load_from_file(str(obj), str(a_png), None, None, None, str(n_jpg), None,
                                                cgmath.mul(cgmath.from_translation((-1.7, 0.5, 2.7)), cgmath.from_angle_y(45.0), cgmath.from_scale(0.4))))
    gpu_ctx.upload(sc.flatten())
    _, u8, _, _ = gpu_ctx.render(sc.camera, seed=1)
    assert np.array_equal(read_ppm(out), u8)
    assert int(u8[40:, :64].max()) > 0                                     # the textured cube is in the picture


@pytest.mark.gpu
def test_cpp_caller_through_mi_multi(gpu_ctx, tmp_path):
    """The compiled caller over mi_multi_* (one device here; RCCL is dlopen'ed and a one-device communicator built):
    same bytes as the single-context call."""
    a, b = tmp_path / "single.ppm", tmp_path / "multi.ppm"
    r1 = subprocess.run([CLI, str(a), "96", "80", "16", "8"], capture_output=True, text=True)
    r2 = subprocess.run([CLI, str(b), "96", "80", "16", "8", "-", "1"], capture_output=True, text=True)
    assert r1.returncode == 0 and r2.returncode == 0, r1.stderr + r2.stderr
    assert "gpus=1" in r2.stdout
    assert np.array_equal(read_ppm(a), read_ppm(b))
