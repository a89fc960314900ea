"""The five BASELINE.json configurations as scenes built ONLY from the reference's
primitive and material types.  The Cornell-box scene code is not in the reference's
HEAD (only pictures of it: renders/render.png, README.md:43,55), so the scenes are
defined here following SURVEY.md §8(d); parameter values not stated by BASELINE.json
come from the reference's run() literal (tracing.rs:357-373): path_depth 10,
max_trace_dist 100, gamma 2, focal_length 0.6, focus_dist 5, path_samples 1.
"""
from __future__ import annotations

import os
import numpy as np

from . import cgmath, objload
from .geometry import ConvexVolume, Plane, Sphere, StaticMesh, Triangle
from .materials import Dielectric, Isotropic, Lambertian, Metal, ParameterizedMaterial
from .texture import Texture
from .tracing import Camera, Scene

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def load_asset_mesh(name: str) -> objload.Mesh:
    """Indexed mesh converted from the reference's obj/<name>.obj by tools/make_assets.py."""
    z = np.load(os.path.join(ASSETS, name + ".npz"), allow_pickle=False)
    return objload.Mesh(z["positions"], z["normals"], z["texcoords"], z["indices"], name)


def load_asset_textures():
    """RGB8 decodes of the reference's texture/{green.png, magenta.jpg, normal_test.jpg, normal_test.png}
    (tools/make_assets.py), keyed as head_scene expects."""
    z = np.load(os.path.join(ASSETS, "textures.npz"), allow_pickle=False)
    return {k: Texture(z[k]) for k in z.files}


def _quad(p0, p1, p2, p3, material):
    """Two `Triangle`s (geometry.rs:424), counter-clockwise seen from inside the box."""
    return [Triangle(p0, p1, p2, material), Triangle(p0, p2, p3, material)]


def cornell_camera(width, height, spp, depth=10, lens_radius=0.0):
    return Camera(eyepoint=(0.0, 3.0, 6.6), view_dir=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0),
                  path_depth=depth, path_samples=1, screen_width=width, screen_height=height,
                  focal_length=0.6, focus_dist=5.0, lens_radius=lens_radius, aa_sample_count=spp,
                  max_trace_dist=100.0, gamma=2.0)


def cornell_walls():
    """5 quads = 10 Triangles: x in [-3,3], y in [0,6], z in [-3,3], open toward +z.
    The ceiling quad is the light (emissive Lambertian, as HEAD lights its scene,
    tracing.rs:527-538)."""
    grey = Lambertian(albedo=(0.73, 0.73, 0.73))
    red = Lambertian(albedo=(0.65, 0.05, 0.05))
    green = Lambertian(albedo=(0.12, 0.45, 0.15))
    light = Lambertian(albedo=(0.73, 0.73, 0.73), emission=(4.0, 4.0, 4.0))
    x0, x1, y0, y1, z0, z1 = -3.0, 3.0, 0.0, 6.0, -3.0, 3.0
    objs = []
    objs += _quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), grey)     # floor  (normal +y)
    objs += _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), grey)     # back   (normal +z)
    objs += _quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), red)      # left   (normal +x)
    objs += _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), green)    # right  (normal -x)
    objs += _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), light)    # ceiling light (normal -y)
    return objs


def cornell_spheres():
    return [
        Sphere((-1.4, 1.0, -0.5), 1.0, Metal(albedo=(0.8, 0.8, 0.8), emission=(0.0, 0.0, 0.0), roughness=0.1)),
        Sphere((1.4, 1.0, 0.8), 1.0, Dielectric(idx_of_refraction=1.5)),
    ]


def teapot_mesh():
    """obj/teapot.obj (z-up) as a StaticMesh with a fixed Lambertian material."""
    xf = cgmath.mul(cgmath.from_translation((0.0, 0.9, 0.0)), cgmath.from_angle_x(-90.0), cgmath.from_scale(2.2))
    return StaticMesh(load_asset_mesh("teapot"), Lambertian(albedo=(0.5, 0.02, 0.5)), [None] * 5, xf)


def config1(width=400, height=400, spp=16, depth=8):
    """configs[0]: Cornell box (5 quads + 2 spheres), 400x400, 16 spp, depth 8."""
    return Scene(cornell_camera(width, height, spp, depth), cornell_walls() + cornell_spheres())


def config2(width=1920, height=1080, spp=256, depth=10, lens_radius=0.0):
    """configs[1] (and configs[2] with spp=1024): Cornell box + Utah teapot BVH."""
    return Scene(cornell_camera(width, height, spp, depth, lens_radius),
                 cornell_walls() + cornell_spheres() + [teapot_mesh()])


def config3(width=1920, height=1080, spp=1024, depth=10):
    return config2(width, height, spp, depth)


def synthetic_drone_maps(size=2048, seed=7):
    """The reference's Drone_*.tga maps are absent (.MISSING_LARGE_BLOBS); deterministic
    procedural stand-ins of the same role: albedo noise, sparse emission, grey-gradient
    metallic / roughness, a tiled tangent-space normal pattern."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / size
    cell = rng.integers(40, 230, size=(size // 32, size // 32, 3), dtype=np.uint8)
    albedo = np.kron(cell, np.ones((32, 32, 1), np.uint8))
    emission = np.zeros((size, size, 3), np.uint8)
    mask = rng.random((size // 64, size // 64)) < 0.03
    emission[np.kron(mask, np.ones((64, 64), bool))] = (255, 160, 40)
    metallic = np.repeat((xx * 255).astype(np.uint8)[:, :, None], 3, axis=2)
    roughness = np.repeat((yy * 200 + 30).astype(np.uint8)[:, :, None], 3, axis=2)
    nx = 0.25 * np.sin(xx * 64 * np.pi)
    ny = 0.25 * np.cos(yy * 64 * np.pi)
    nz = np.sqrt(np.clip(1.0 - nx * nx - ny * ny, 0, 1))
    normal = (np.stack([nx, ny, nz], axis=2) * 0.5 + 0.5)
    normal = (normal * 255).astype(np.uint8)
    return [Texture(t) for t in (albedo, emission, metallic, roughness, normal)]


def drone_mesh(maps=None, size=2048):
    """obj/drone.obj with HEAD's transform (tracing.rs:383) scaled up to sit in the box."""
    maps = synthetic_drone_maps(size) if maps is None else maps
    xf = cgmath.mul(cgmath.from_translation((0.0, 2.2, 0.6)), cgmath.from_angle_y(-60.0),
                    cgmath.from_angle_x(180.0), cgmath.from_scale(0.0075))
    return StaticMesh(load_asset_mesh("drone"), None, maps, xf)


def config4(width=1920, height=1080, spp=256, depth=10, tex_size=2048):
    """configs[3]: textured drone mesh (albedo/normal/metallic/roughness/emission maps) in the Cornell box."""
    return Scene(cornell_camera(width, height, spp, depth), cornell_walls() + [drone_mesh(size=tex_size)])


def config5(width=1920, height=1080, spp=4096, depth=50):
    """configs[4]: dielectric sphere enclosing an isotropic volume ("subsurface", README.md:68-69)
    + glass + metal, depth 50."""
    skin = Sphere((0.0, 1.3, 0.3), 1.3, Dielectric(idx_of_refraction=1.5))
    inner = ConvexVolume(Sphere((0.0, 1.3, 0.3), 1.3, Dielectric(idx_of_refraction=1.5)),
                         Isotropic(albedo=(0.9, 0.6, 0.5), emission=(0.0, 0.0, 0.0)), 4.0)
    glass = Sphere((2.0, 0.7, 1.4), 0.7, Dielectric(idx_of_refraction=1.5))
    metal = Sphere((-2.0, 0.8, 1.0), 0.8, Metal(albedo=(0.8, 0.8, 0.8), emission=(0.0, 0.0, 0.0), roughness=0.1))
    return Scene(cornell_camera(width, height, spp, depth), cornell_walls() + [skin, inner, glass, metal])


def furnace(albedo=0.5, emission=1.0, depth=6, size=32, spp=64):
    """Closed emissive Lambertian box for the analytic furnace identity:
    radiance = E * sum_{k<depth} (0.75 a)^k at every pixel."""
    wall = Lambertian(albedo=(albedo,) * 3, emission=(emission,) * 3)
    x0, x1, y0, y1, z0, z1 = -1.0, 1.0, -1.0, 1.0, -1.0, 1.0
    objs = []
    objs += _quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), wall)
    objs += _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), wall)
    objs += _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), wall)
    objs += _quad((x1, y0, z1), (x0, y0, z1), (x0, y1, z1), (x1, y1, z1), wall)
    objs += _quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), wall)
    objs += _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), wall)
    cam = Camera(eyepoint=(0.0, 0.0, 0.0), view_dir=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), path_depth=depth,
                 path_samples=1, screen_width=size, screen_height=size, focal_length=0.6, focus_dist=5.0,
                 lens_radius=0.0, aa_sample_count=spp, max_trace_dist=100.0, gamma=2.0)
    return Scene(cam, objs)


def head_scene(width=100, height=100, spp=100, depth=10, with_meshes=True, textures=None):
    """The reference's own run() scene (tracing.rs:356-543).  The drone's TGA maps are
    missing from the reference (.MISSING_LARGE_BLOBS) -> Texture::load_from_file gives
    None for each, exactly as the reference would today (texture.rs:22-24).
    `textures` may supply decoded stand-ins {name: Texture} for green/normal_test/magenta."""
    textures = textures or {}
    objs = []
    if with_meshes:
        objs.append(StaticMesh(load_asset_mesh("drone"), None, [None] * 5,
                               cgmath.mul(cgmath.from_translation((0.0, 1.3, 1.7)), cgmath.from_angle_y(-60.0),
                                          cgmath.from_angle_x(180.0), cgmath.from_scale(0.0030))))
        objs.append(StaticMesh(load_asset_mesh("cube"), None,
                               [textures.get("green"), None, None, None, textures.get("normal_test_jpg")],
                               cgmath.mul(cgmath.from_translation((-1.7, 0.5, 2.7)), cgmath.from_angle_y(45.0),
                                          cgmath.from_scale(0.4))))
        objs.append(StaticMesh(load_asset_mesh("sphere"), None,
                               [textures.get("magenta"), None, None, None, textures.get("normal_test_png")],
                               cgmath.mul(cgmath.from_translation((1.7, 0.5, 2.7)), cgmath.from_angle_y(45.0),
                                          cgmath.from_scale(0.6))))
    for row, (y, metallic) in enumerate(((3.3, 0.0), (4.4, 0.5), (5.5, 1.0))):
        for x, rough in ((-2.6, 0.0), (-1.3, 0.25), (0.0, 0.5), (1.3, 0.75), (2.6, 1.0)):
            objs.append(Sphere((x, y, 0.0), 0.5, ParameterizedMaterial(albedo=(0.01, 0.02, 0.5), emission=(0, 0, 0),
                                                                       roughness=rough, metallic=metallic)))
    objs.append(Sphere((-2.3, 2.0, 2.0), 0.4, Dielectric(idx_of_refraction=2.5)))
    objs.append(Sphere((2.3, 2.0, 2.0), 0.4, Lambertian(albedo=(0.3, 0.3, 0.3), emission=(0.0, 1.0, 1.0))))
    objs.append(ConvexVolume(Sphere((-3.0, 1.0, 1.0), 1.0, Dielectric(1.5)),
                             Isotropic(albedo=(1.0, 1.0, 1.0), emission=(0, 0, 0)), 0.6))
    objs.append(ConvexVolume(Sphere((3.0, 1.0, 1.0), 1.0, Dielectric(1.5)),
                             Isotropic(albedo=(0.0, 0.0, 0.0), emission=(0, 0, 0)), 0.8))
    objs.append(Plane((0.0, 0.0, 0.0), (0.0, 1.0, 0.0),
                      ParameterizedMaterial(albedo=(0.33, 0.33, 0.33), emission=(0, 0, 0), metallic=0.3, roughness=0.7)))
    light = Lambertian(albedo=(0.0, 0.6, 0.0), emission=(7.0, 7.0, 7.0))
    objs.append(Triangle((-2.5, 7.5, -0.5), (2.5, 7.5, -0.5), (2.5, 7.5, 3.5), light))
    objs.append(Triangle((-2.5, 7.5, -0.5), (-2.5, 7.5, 3.5), (2.5, 7.5, 3.5), light))
    cam = Camera(eyepoint=(0.0, 2.0, 5.5), view_dir=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), path_depth=depth,
                 path_samples=1, screen_width=width, screen_height=height, focal_length=0.6, focus_dist=5.0,
                 lens_radius=0.0, aa_sample_count=spp, max_trace_dist=100.0, gamma=2.0)
    return Scene(cam, objs)


def with_camera(scene, point_light_pos=None, ambient=None, **camera_fields):
    """Same scene with some Camera fields replaced (projection_mode, shading_mode, ...) and, for
    ShadingMode::Phong, the Scene's point light and ambient term (tracing.rs:216-217)."""
    for k, v in camera_fields.items():
        if not hasattr(scene.camera, k):
            raise AttributeError(k)
        setattr(scene.camera, k, v)
    if point_light_pos is not None:
        scene.point_light_pos = tuple(point_light_pos)
    if ambient is not None:
        scene.ambient = tuple(ambient)
    return scene
