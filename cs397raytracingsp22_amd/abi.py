"""ctypes binding of the C ABI declared in include/mi_rt.h (libmi_rt.so).

This is plumbing only: it mirrors the PODs of mi_rt.h one to one and loads the HIP
library.  There is NO CPU fallback anywhere in this package: if the library is
missing, `load()` raises, and every entry point that needs the GPU fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi_rt.so")

MI_TILE = 32

# status codes (mi_status)
MI_RT_ABI_VERSION = 5      # include/mi_rt.h
MI_OK, MI_ERR_INVALID, MI_ERR_UNSUPPORTED, MI_ERR_NO_DEVICE, MI_ERR_HIP, MI_ERR_OOM, MI_ERR_NO_SCENE = 0, -1, -2, -3, -4, -5, -6
# material kinds
MI_MAT_LAMBERTIAN, MI_MAT_METAL, MI_MAT_DIELECTRIC, MI_MAT_PARAMETERIZED, MI_MAT_ISOTROPIC = range(5)
# object kinds
MI_OBJ_SPHERE, MI_OBJ_TRIANGLE, MI_OBJ_PLANE, MI_OBJ_VOLUME, MI_OBJ_MESH, MI_OBJ_SCENE = range(6)
MI_PROJ_ORTHOGRAPHIC, MI_PROJ_PERSPECTIVE = 0, 1
MI_SHADE_PHONG, MI_SHADE_PATHTRACE = 0, 1
MI_VARIANT_DEFAULT, MI_VARIANT_SIMPLE, MI_VARIANT_VOTED, MI_VARIANT_VOTED_DIAG = 0, 1, 3, 4      # 2, 5, 6: removed in ABI 3
MI_VARIANT_WAVEFRONT, MI_VARIANT_RECURSIVE = 7, 8
MI_OPT_NO_TILE_MASKS, MI_OPT_REFERENCE_WALK, MI_OPT_TWO_STAGE, MI_OPT_NO_LIST_TREE = 1, 2, 4, 8

f3 = C.c_float * 3
f16 = C.c_float * 16
i5 = C.c_int32 * 5


class mi_material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", f3), ("emission", f3), ("roughness", C.c_float),
                ("metallic", C.c_float), ("idx_of_refraction", C.c_float)]


class mi_object(C.Structure):
    _fields_ = [("kind", C.c_int32), ("index", C.c_int32)]


class mi_sphere(C.Structure):
    _fields_ = [("center", f3), ("radius", C.c_float), ("material", C.c_int32)]


class mi_triangle(C.Structure):
    _fields_ = [("a", f3), ("b", f3), ("c", f3), ("material", C.c_int32)]


class mi_plane(C.Structure):
    _fields_ = [("point", f3), ("normal", f3), ("material", C.c_int32)]


class mi_volume(C.Structure):
    _fields_ = [("boundary_center", f3), ("boundary_radius", C.c_float), ("density", C.c_float),
                ("phase_material", C.c_int32), ("boundary_kind", C.c_int32), ("boundary_index", C.c_int32),
                ("boundary_count", C.c_int32)]


class mi_texture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgb", C.POINTER(C.c_uint8))]


class mi_mesh(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)),
                ("texcoords", C.POINTER(C.c_float)), ("indices", C.POINTER(C.c_uint32)),
                ("n_vertices", C.c_int32), ("n_triangles", C.c_int32),
                ("transform", f16), ("inv_transform", f16),
                ("material", C.c_int32), ("textures", i5)]


class mi_scene_desc(C.Structure):
    _fields_ = [("objects", C.POINTER(mi_object)), ("n_objects", C.c_int32),
                ("spheres", C.POINTER(mi_sphere)), ("n_spheres", C.c_int32),
                ("triangles", C.POINTER(mi_triangle)), ("n_triangles", C.c_int32),
                ("planes", C.POINTER(mi_plane)), ("n_planes", C.c_int32),
                ("volumes", C.POINTER(mi_volume)), ("n_volumes", C.c_int32),
                ("meshes", C.POINTER(mi_mesh)), ("n_meshes", C.c_int32),
                ("materials", C.POINTER(mi_material)), ("n_materials", C.c_int32),
                ("textures", C.POINTER(mi_texture)), ("n_textures", C.c_int32),
                ("boundary_objects", C.POINTER(mi_object)), ("n_boundary_objects", C.c_int32),
                ("point_light_pos", f3), ("ambient", f3)]


class mi_camera_desc(C.Structure):
    _fields_ = [("eyepoint", f3), ("view_dir", f3), ("up", f3),
                ("projection_mode", C.c_int32), ("shading_mode", C.c_int32),
                ("path_depth", C.c_uint32), ("path_samples", C.c_uint32),
                ("screen_width", C.c_uint32), ("screen_height", C.c_uint32),
                ("focal_length", C.c_float), ("focus_dist", C.c_float), ("lens_radius", C.c_float),
                ("aa_sample_count", C.c_uint32), ("max_trace_dist", C.c_float), ("gamma", C.c_float)]


class mi_render_opts(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("rank", C.c_int32), ("world", C.c_int32),
                ("variant", C.c_int32), ("want_signature", C.c_int32), ("flags", C.c_uint32),
                ("max_state_bytes", C.c_uint64)]


class mi_stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("pixels", C.c_uint64), ("tiles", C.c_uint32),
                ("tiles_padded", C.c_uint32), ("kernel_ms", C.c_float), ("total_ms", C.c_float),
                ("scene_bytes", C.c_uint32), ("scene_in_lds", C.c_uint32)]


# every symbol include/mi_rt.h declares (tests check the library exports exactly these)
EXPORTS = [
    "mi_ctx_create", "mi_ctx_destroy", "mi_scene_upload", "mi_render", "mi_compact_size",
    "mi_render_tiles_device", "mi_unpermute_device", "mi_tonemap_device", "mi_last_kernel_ms",
    "mi_reserve", "mi_render_samples_device", "mi_last_pipeline_ms", "mi_last_pipeline_counts", "mi_last_diag", "mi_selftest", "mi_last_error", "mi_abi_version",
    "mi_multi_create", "mi_multi_create_loopback", "mi_multi_destroy", "mi_multi_device_count", "mi_multi_context", "mi_multi_scene_upload", "mi_multi_reserve", "mi_multi_render",
]

_lib = None


class MiError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mi_rt error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load libmi_rt.so (built in-tree by __graft_entry__.build()).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP library has not been built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for the render path.")
    # If the process also uses torch (device tensors, torch.distributed), torch's bundled HIP
    # runtime must be the one in the process: load it before libmi_rt.so resolves libamdhip64.
    import sys
    if "torch" in sys.modules or os.environ.get("MI_RT_PRELOAD_TORCH") == "1":
        import torch  # noqa: F401
        if torch.cuda.is_available():
            torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.mi_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.mi_ctx_create.restype = C.c_int
    lib.mi_ctx_destroy.argtypes = [vp]
    lib.mi_ctx_destroy.restype = None
    lib.mi_scene_upload.argtypes = [vp, C.POINTER(mi_scene_desc)]
    lib.mi_scene_upload.restype = C.c_int
    lib.mi_render.argtypes = [vp, C.POINTER(mi_camera_desc), C.POINTER(mi_render_opts),
                              vp, vp, vp, C.POINTER(mi_stats)]
    lib.mi_render.restype = C.c_int
    lib.mi_compact_size.argtypes = [C.POINTER(mi_camera_desc), C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.mi_compact_size.restype = C.c_int
    lib.mi_render_tiles_device.argtypes = [vp, C.POINTER(mi_camera_desc), C.POINTER(mi_render_opts),
                                           vp, vp, vp, C.POINTER(mi_stats)]
    lib.mi_render_tiles_device.restype = C.c_int
    lib.mi_render_samples_device.argtypes = [vp, C.POINTER(mi_camera_desc), C.POINTER(mi_render_opts), C.c_uint32, C.c_uint32,
                                             vp, vp, vp, vp, C.POINTER(mi_stats)]
    lib.mi_render_samples_device.restype = C.c_int
    lib.mi_unpermute_device.argtypes = [vp, C.POINTER(mi_camera_desc), C.c_int32, vp, vp, vp]
    lib.mi_unpermute_device.restype = C.c_int
    lib.mi_tonemap_device.argtypes = [vp, C.POINTER(mi_camera_desc), vp, vp, vp]
    lib.mi_tonemap_device.restype = C.c_int
    lib.mi_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.mi_last_kernel_ms.restype = C.c_int
    lib.mi_reserve.argtypes = [vp, C.POINTER(mi_camera_desc), C.c_int32, C.c_uint64]
    lib.mi_reserve.restype = C.c_int
    lib.mi_last_pipeline_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.mi_last_pipeline_ms.restype = C.c_int
    lib.mi_last_pipeline_counts.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.mi_last_pipeline_counts.restype = C.c_int
    lib.mi_last_diag.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.mi_last_diag.restype = C.c_int
    lib.mi_selftest.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.mi_selftest.restype = C.c_int
    lib.mi_multi_create.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    lib.mi_multi_create.restype = C.c_int
    lib.mi_multi_create_loopback.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
    lib.mi_multi_create_loopback.restype = C.c_int
    lib.mi_multi_destroy.argtypes = [vp]
    lib.mi_multi_destroy.restype = None
    lib.mi_multi_device_count.argtypes = [vp]
    lib.mi_multi_device_count.restype = C.c_int
    lib.mi_multi_context.argtypes = [vp, C.c_int]
    lib.mi_multi_context.restype = vp
    lib.mi_multi_scene_upload.argtypes = [vp, C.POINTER(mi_scene_desc)]
    lib.mi_multi_scene_upload.restype = C.c_int
    lib.mi_multi_reserve.argtypes = [vp, C.POINTER(mi_camera_desc), C.c_uint64]
    lib.mi_multi_reserve.restype = C.c_int
    lib.mi_multi_render.argtypes = [vp, C.POINTER(mi_camera_desc), C.POINTER(mi_render_opts), vp, vp, vp, C.POINTER(mi_stats)]
    lib.mi_multi_render.restype = C.c_int
    lib.mi_last_error.argtypes = []
    lib.mi_last_error.restype = C.c_char_p
    lib.mi_abi_version.argtypes = []
    lib.mi_abi_version.restype = C.c_int
    _lib = lib
    return lib


def check(code: int) -> None:
    if code != MI_OK:
        msg = load().mi_last_error()
        raise MiError(code, msg.decode("utf-8", "replace") if msg else "")
