"""cs397raytracingsp22_amd — MI355X-native path-tracing hot path behind the reference's
Scene / Camera / Material / Intersectable surface (mbk6/CS397RayTracingSP22).

Layout:
  csrc/        HIP kernels (gfx950) + the C-ABI host runtime  -> lib/libmi_rt.so
  abi.py       ctypes binding of include/mi_rt.h
  tracing.py, geometry.py, materials.py, texture.py
               host-side mirror of the reference modules of the same names
  scenes.py    the BASELINE.json configurations
  dist.py      one-process-per-GPU tile sharding + the single RCCL gather

The render path exists only as HIP code: importing this package never falls back to a
CPU implementation, and abi.load() raises when lib/libmi_rt.so has not been built.
"""
from . import abi  # noqa: F401
from .materials import Dielectric, Isotropic, Lambertian, Material, Metal, ParameterizedMaterial  # noqa: F401
from .geometry import ConvexVolume, Intersectable, Plane, Sphere, StaticMesh, Triangle  # noqa: F401
from .texture import Texture  # noqa: F401
from .tracing import Camera, CameraProjectionMode, Context, MultiContext, Scene, ShadingMode, compact_size  # noqa: F401

__all__ = [
    "abi", "Camera", "CameraProjectionMode", "ShadingMode", "Scene", "Context", "MultiContext", "compact_size",
    "Intersectable", "Sphere", "Triangle", "Plane", "ConvexVolume", "StaticMesh",
    "Material", "Lambertian", "Metal", "Dielectric", "ParameterizedMaterial", "Isotropic", "Texture",
]
