"""Host-side mirror of src/util/tracing.rs: `Camera` and `Scene` with the reference's
field names; `Scene.render_to_image()` keeps its meaning (tracing.rs:221-263) and
becomes flatten -> one call through the C ABI (include/mi_rt.h) -> image bytes.

Everything below `render_to_image` — generate_rays, shade_ray, the hit loop, every
intersect_ray / scatter / texture sample — runs in the HIP kernels of csrc/.  There is
no CPU implementation of the path in this package.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional
import numpy as np

from . import abi
from .geometry import FlatBuilder, FlatScene, Intersectable


class CameraProjectionMode:          # tracing.rs:27-30
    Orthographic = abi.MI_PROJ_ORTHOGRAPHIC
    Perspective = abi.MI_PROJ_PERSPECTIVE


class ShadingMode:                   # tracing.rs:32-35
    Phong = abi.MI_SHADE_PHONG
    PathTrace = abi.MI_SHADE_PATHTRACE


@dataclass
class Camera:                        # tracing.rs:138-155, same fields
    eyepoint: tuple = (0.0, 2.0, 5.5)
    view_dir: tuple = (0.0, 0.0, -1.0)
    up: tuple = (0.0, 1.0, 0.0)
    projection_mode: int = CameraProjectionMode.Perspective
    shading_mode: int = ShadingMode.PathTrace
    path_depth: int = 10
    path_samples: int = 1
    screen_width: int = 100
    screen_height: int = 100
    focal_length: float = 0.6
    focus_dist: float = 5.0
    lens_radius: float = 0.0
    aa_sample_count: int = 100
    max_trace_dist: float = 100.0
    gamma: float = 2.0

    def to_pod(self) -> abi.mi_camera_desc:
        c = abi.mi_camera_desc()
        c.eyepoint = abi.f3(*np.asarray(self.eyepoint, np.float32))
        c.view_dir = abi.f3(*np.asarray(self.view_dir, np.float32))
        c.up = abi.f3(*np.asarray(self.up, np.float32))
        c.projection_mode, c.shading_mode = self.projection_mode, self.shading_mode
        c.path_depth, c.path_samples = self.path_depth, self.path_samples
        c.screen_width, c.screen_height = self.screen_width, self.screen_height
        c.focal_length, c.focus_dist, c.lens_radius = self.focal_length, self.focus_dist, self.lens_radius
        c.aa_sample_count = self.aa_sample_count
        c.max_trace_dist, c.gamma = self.max_trace_dist, self.gamma
        return c


class Context:
    """One mi_ctx = one GPU (one process per GPU: pass LOCAL_RANK)."""

    def __init__(self, device: int = 0, _borrowed=None):
        self._lib = abi.load()
        self._h = C.c_void_p()
        self._owned = _borrowed is None
        if _borrowed is None:
            abi.check(self._lib.mi_ctx_create(device, C.byref(self._h)))
        else:                                  # a device context owned by a MultiContext (mi_multi_context)
            self._h = C.c_void_p(_borrowed)
        self.device = device

    def close(self):
        if self._h and self._owned:
            self._lib.mi_ctx_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, flat: FlatScene):
        abi.check(self._lib.mi_scene_upload(self._h, C.byref(flat.desc)))

    def render(self, cam: Camera, seed: int = 1, want_f32=True, want_u8=True, want_sig=False,
               variant: int = abi.MI_VARIANT_DEFAULT, flags: int = 0, max_state_bytes: int = 0):
        """mi_render: whole image on this GPU.  Returns (f32 [H,W,3] | None, u8 [H,W,3] | None,
        sig [H,W] | None, mi_stats)."""
        pod = cam.to_pod()
        opts = abi.mi_render_opts(seed=seed, rank=0, world=1, variant=variant, want_signature=int(want_sig),
                                  flags=flags, max_state_bytes=max_state_bytes)
        H, W = cam.screen_height, cam.screen_width
        f32 = np.empty((H, W, 3), np.float32) if want_f32 else None
        u8 = np.empty((H, W, 3), np.uint8) if want_u8 else None
        sig = np.empty((H, W), np.uint32) if want_sig else None
        st = abi.mi_stats()
        abi.check(self._lib.mi_render(
            self._h, C.byref(pod), C.byref(opts),
            f32.ctypes.data if f32 is not None else None,
            u8.ctypes.data if u8 is not None else None,
            sig.ctypes.data if sig is not None else None, C.byref(st)))
        return f32, u8, sig, st

    # ---- device-pointer building blocks (multi-GPU; pointers are ints, e.g. tensor.data_ptr()) ----
    def render_tiles_device(self, cam: Camera, d_compact: int, d_sig: Optional[int] = None, seed: int = 1,
                            rank: int = 0, world: int = 1, stream: Optional[int] = None,
                            variant: int = abi.MI_VARIANT_DEFAULT, flags: int = 0, max_state_bytes: int = 0):
        pod = cam.to_pod()
        opts = abi.mi_render_opts(seed=seed, rank=rank, world=world, variant=variant,
                                  want_signature=int(d_sig is not None), flags=flags, max_state_bytes=max_state_bytes)
        st = abi.mi_stats()
        abi.check(self._lib.mi_render_tiles_device(self._h, C.byref(pod), C.byref(opts), d_compact, d_sig,
                                                   stream, C.byref(st)))
        return st

    def render_samples_device(self, cam: Camera, sample_begin: int, sample_end: int, d_accum: int,
                              d_compact: Optional[int] = None, d_sig: Optional[int] = None, seed: int = 1,
                              rank: int = 0, world: int = 1, stream: Optional[int] = None,
                              flags: int = 0, max_state_bytes: int = 0):
        """mi_render_samples_device: add samples [sample_begin, sample_end) of every pixel, in order, to the
        caller-held accumulator (float4 per compact pixel).  The call that reaches aa_sample_count also
        writes the means to d_compact.  Progressive display and checkpoint / resume are built on this."""
        pod = cam.to_pod()
        opts = abi.mi_render_opts(seed=seed, rank=rank, world=world, variant=abi.MI_VARIANT_DEFAULT,
                                  want_signature=int(d_sig is not None), flags=flags, max_state_bytes=max_state_bytes)
        st = abi.mi_stats()
        abi.check(self._lib.mi_render_samples_device(self._h, C.byref(pod), C.byref(opts), sample_begin, sample_end,
                                                     d_accum, d_compact, d_sig, stream, C.byref(st)))
        return st

    def unpermute_device(self, cam: Camera, world: int, d_gathered: int, d_image: int, stream: Optional[int] = None):
        pod = cam.to_pod()
        abi.check(self._lib.mi_unpermute_device(self._h, C.byref(pod), world, d_gathered, d_image, stream))

    def tonemap_device(self, cam: Camera, d_image_f32: int, d_image_u8: int, stream: Optional[int] = None):
        pod = cam.to_pod()
        abi.check(self._lib.mi_tonemap_device(self._h, C.byref(pod), d_image_f32, d_image_u8, stream))

    def reserve(self, cam: Camera, world: int = 1, max_state_bytes: int = 0):
        """Allocate the wavefront pipeline's HBM buffers ahead of the first render (mi_reserve)."""
        pod = cam.to_pod()
        abi.check(self._lib.mi_reserve(self._h, C.byref(pod), world, max_state_bytes))

    def last_pipeline_ms(self):
        """Wavefront pipeline of the last render: dict of per-kernel duration sums (ms) and launch count."""
        out = (C.c_float * 8)()
        abi.check(self._lib.mi_last_pipeline_ms(self._h, out))
        return {"wf_main_ms": float(out[0]), "wf_trav_ms": float(out[1]), "wf_reduce_ms": float(out[2]), "launches": int(out[3]),
                "wf_trav_f_ms": float(out[4]), "wf_replay_ms": float(out[5]), "wf_main_a_ms": float(out[6])}

    def last_pipeline_counts(self):
        """Path counts of the last wavefront render (mi_last_pipeline_counts), for traffic accounting."""
        out = (C.c_uint64 * 8)()
        abi.check(self._lib.mi_last_pipeline_counts(self._h, out))
        k = ["passes", "paths_a", "paths_b", "queue_entries", "sample_slots", "pixels", "segments", "dead_tile_samples"]
        return dict(zip(k, [int(v) for v in out]))

    def last_diag(self):
        """Counters of the last MI_VARIANT_VOTED_DIAG launch as a dict (diagnostic)."""
        out = (C.c_uint64 * 16)()
        abi.check(self._lib.mi_last_diag(self._h, out))
        k = ["a_trips", "a_lanes", "inner_trips", "inner_lanes", "leaf_trips", "leaf_lanes", "b_trips", "waves",
             "a_cycles", "b_cycles", "shade_cycles", "gen_cycles", "list_cycles", "slab_tests", "segments"]
        return dict(zip(k, [int(v) for v in out]))

    def selftest(self):
        """mi_selftest: exhaustive check of the kernels' short reciprocal against the IEEE division -> (mismatches, checked)."""
        out = (C.c_uint64 * 4)()
        abi.check(self._lib.mi_selftest(self._h, out))
        return int(out[0]), int(out[1])

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        abi.check(self._lib.mi_last_kernel_ms(self._h, C.byref(ms)))
        return float(ms.value)


class MultiContext:
    """mi_multi: N GPUs of one node behind ONE handle (one context, stream and host thread per device inside the
    library; tiles t % N; a single RCCL send/recv fan-in per frame; un-permute and tone-map on device 0)."""

    def __init__(self, n_devices: int, devices=None, _loopback_device=None):
        self._lib = abi.load()
        self._h = C.c_void_p()
        if _loopback_device is not None:
            abi.check(self._lib.mi_multi_create_loopback(n_devices, _loopback_device, C.byref(self._h)))
        else:
            arr = (C.c_int * n_devices)(*devices) if devices is not None else None
            abi.check(self._lib.mi_multi_create(n_devices, arr, C.byref(self._h)))
        self.n_devices = n_devices

    @classmethod
    def loopback(cls, n_contexts: int, device: int = 0) -> "MultiContext":
        """mi_multi_create_loopback: the TEST transport — `n_contexts` ranks on one device, the RCCL fan-in replaced by
        event-ordered device-to-device copies; everything else is mi_multi_render's N >= 2 code."""
        return cls(n_contexts, _loopback_device=device)

    def close(self):
        if self._h:
            self._lib.mi_multi_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, flat: FlatScene):
        abi.check(self._lib.mi_multi_scene_upload(self._h, C.byref(flat.desc)))

    def context(self, rank: int) -> Context:
        """The context of device number `rank`, borrowed (mi_multi_context): per-device timing and path counts."""
        h = self._lib.mi_multi_context(self._h, rank)
        if not h:
            abi.check(abi.MI_ERR_INVALID)
        return Context(rank, _borrowed=h)

    def reserve(self, cam: Camera, max_state_bytes: int = 0):
        pod = cam.to_pod()
        abi.check(self._lib.mi_multi_reserve(self._h, C.byref(pod), max_state_bytes))

    def render(self, cam: Camera, seed: int = 1, want_f32=True, want_u8=True, want_sig=False, flags: int = 0,
               max_state_bytes: int = 0, variant: int = abi.MI_VARIANT_DEFAULT):
        """mi_multi_render: the whole image, assembled on device 0.  Same return value as Context.render.  With no output
        wanted the finished f32 and u8 images stay resident on device 0 (nothing crosses PCIe)."""
        pod = cam.to_pod()
        opts = abi.mi_render_opts(seed=seed, rank=0, world=1, variant=variant, want_signature=int(want_sig),
                                  flags=flags, max_state_bytes=max_state_bytes)
        H, W = cam.screen_height, cam.screen_width
        f32 = np.empty((H, W, 3), np.float32) if want_f32 else None
        u8 = np.empty((H, W, 3), np.uint8) if want_u8 else None
        sig = np.empty((H, W), np.uint32) if want_sig else None
        st = abi.mi_stats()
        abi.check(self._lib.mi_multi_render(
            self._h, C.byref(pod), C.byref(opts),
            f32.ctypes.data if f32 is not None else None,
            u8.ctypes.data if u8 is not None else None,
            sig.ctypes.data if sig is not None else None, C.byref(st)))
        return f32, u8, sig, st


def compact_size(cam: Camera, world: int):
    """(tiles_total, tiles_padded) of the tile partition (mi_compact_size)."""
    pod = cam.to_pod()
    a, b = C.c_uint32(), C.c_uint32()
    abi.check(abi.load().mi_compact_size(C.byref(pod), world, C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


@dataclass
class Scene:                         # tracing.rs:213-218
    camera: Camera
    objects: List[Intersectable]
    point_light_pos: tuple = (0.0, 1.0, 5.0)    # read by ShadingMode::Phong only (tracing.rs:282,288)
    ambient: tuple = (0.1, 0.1, 0.1)            # Phong only (:292)

    def flatten(self) -> FlatScene:
        fb = FlatBuilder()
        for obj in self.objects:
            obj.flatten(fb)
        flat = fb.finish()
        flat.desc.point_light_pos = abi.f3(*[float(v) for v in self.point_light_pos])
        flat.desc.ambient = abi.f3(*[float(v) for v in self.ambient])
        return flat

    def render_to_image(self, seed: int = 1, device: int = 0) -> np.ndarray:
        """Scene::render_to_image (tracing.rs:221-263): returns the RgbImage bytes [H,W,3] u8."""
        ctx = Context(device)
        try:
            ctx.upload(self.flatten())
            _, u8, _, _ = ctx.render(self.camera, seed=seed, want_f32=False, want_u8=True)
            return u8
        finally:
            ctx.close()
