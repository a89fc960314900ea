"""Host-side mirror of src/util/geometry.rs: the `Intersectable` implementors as value
types with the reference's names and fields.  `intersect_ray` runs on the GPU; these
classes flatten themselves into the PODs of include/mi_rt.h."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional
import numpy as np

from . import abi, cgmath, objload
from .materials import Material
from .texture import Texture


class Intersectable:
    """trait Intersectable (tracing.rs:42-47).  The additive method is `flatten`."""

    def flatten(self, fb: "FlatBuilder") -> None:
        raise NotImplementedError


@dataclass
class Sphere(Intersectable):         # geometry.rs:389-393
    center: tuple
    radius: float
    material: Material

    def flatten(self, fb):
        s = abi.mi_sphere()
        s.center = abi.f3(*np.asarray(self.center, np.float32))
        s.radius = float(self.radius)
        s.material = fb.material(self.material)
        fb.add(abi.MI_OBJ_SPHERE, fb.spheres, s)


@dataclass
class Triangle(Intersectable):       # geometry.rs:424-429
    a: tuple
    b: tuple
    c: tuple
    material: Material

    def flatten(self, fb):
        t = abi.mi_triangle()
        t.a = abi.f3(*np.asarray(self.a, np.float32))
        t.b = abi.f3(*np.asarray(self.b, np.float32))
        t.c = abi.f3(*np.asarray(self.c, np.float32))
        t.material = fb.material(self.material)
        fb.add(abi.MI_OBJ_TRIANGLE, fb.triangles, t)


@dataclass
class Plane(Intersectable):          # geometry.rs:468-472
    point: tuple
    normal: tuple
    material: Material

    def flatten(self, fb):
        p = abi.mi_plane()
        p.point = abi.f3(*np.asarray(self.point, np.float32))
        p.normal = abi.f3(*np.asarray(self.normal, np.float32))
        p.material = fb.material(self.material)
        fb.add(abi.MI_OBJ_PLANE, fb.planes, p)


@dataclass
class ConvexVolume(Intersectable):   # geometry.rs:495-500
    boundary: Intersectable
    phase_function: Material
    density: float

    def flatten(self, fb):
        """`boundary` is `Arc<dyn Intersectable>` (geometry.rs:496): a Sphere travels inline (what every use in the reference
        is), a Triangle / Plane / StaticMesh as an entry of its typed array that is NOT listed in Scene.objects, a nested Scene
        (tracing.rs:326) as a run of such entries."""
        v = abi.mi_volume()
        b = self.boundary
        if isinstance(b, Sphere):
            v.boundary_kind = abi.MI_OBJ_SPHERE
            v.boundary_center = abi.f3(*np.asarray(b.center, np.float32))
            v.boundary_radius = float(b.radius)
        elif isinstance(b, (Triangle, Plane, StaticMesh)):
            v.boundary_kind, v.boundary_index = fb.detached(b)
        elif hasattr(b, "objects"):                                   # a Scene
            entries = []
            for obj in b.objects:
                if not isinstance(obj, (Sphere, Triangle, Plane, StaticMesh)):
                    raise abi.MiError(abi.MI_ERR_UNSUPPORTED, "a ConvexVolume or a Scene inside a ConvexVolume boundary")
                entries.append(fb.detached(obj))
            v.boundary_kind, v.boundary_index, v.boundary_count = abi.MI_OBJ_SCENE, len(fb.boundary_objects), len(entries)
            for kind, index in entries:
                o = abi.mi_object()
                o.kind, o.index = kind, index
                fb.boundary_objects.append(o)
        else:
            raise abi.MiError(abi.MI_ERR_UNSUPPORTED, "ConvexVolume.boundary: a ConvexVolume inside a boundary is not supported")
        v.density = float(self.density)
        v.phase_material = fb.material(self.phase_function)
        fb.add(abi.MI_OBJ_VOLUME, fb.volumes, v)


class StaticMesh(Intersectable):     # geometry.rs:127-134
    def __init__(self, mesh: objload.Mesh, material: Optional[Material], textures, transform):
        self.mesh = mesh
        self.material = material
        self.textures = list(textures)            # 0 albedo 1 emission 2 metallic 3 roughness 4 normal
        self.transform = np.asarray(transform, dtype=np.float32).reshape(4, 4)
        self.inv_transform = cgmath.inverse_transform(self.transform)     # geometry.rs:168

    @staticmethod
    def load_from_file(file_name, albedo_path=None, emission_path=None, metallic_path=None,
                       roughness_path=None, normal_path=None, material=None, transform=None):
        """StaticMesh::load_from_file (geometry.rs:138-172), same argument order.  The BVH
        (geometry.rs:170) is built inside mi_scene_upload."""
        models = objload.load_obj(file_name)
        assert models, "Failed to load OBJ file"                          # geometry.rs:149-150
        texs = [Texture.load_from_file(p) if p is not None else None
                for p in (albedo_path, emission_path, metallic_path, roughness_path, normal_path)]
        return StaticMesh(models[0], material, texs, cgmath.identity() if transform is None else transform)

    def flatten(self, fb):
        if id(self) in fb._mesh_ids:          # the same StaticMesh again (Arc sharing, tracing.rs:215): one mi_mesh, another Scene.objects entry
            o = abi.mi_object()
            o.kind, o.index = abi.MI_OBJ_MESH, fb._mesh_ids[id(self)]
            fb.objects.append(o)
            return
        fb._mesh_ids[id(self)] = len(fb.meshes)
        fb.keep.append(self)
        m = abi.mi_mesh()
        me = self.mesh
        if me.normals.size != me.positions.size or me.texcoords.size * 3 != me.positions.size * 2:
            raise abi.MiError(abi.MI_ERR_INVALID, "mesh needs per-vertex normals and texcoords (geometry.rs:350,355)")
        fb.keep.extend([me.positions, me.normals, me.texcoords, me.indices])
        m.positions = me.positions.ctypes.data_as(abi.C.POINTER(abi.C.c_float))
        m.normals = me.normals.ctypes.data_as(abi.C.POINTER(abi.C.c_float))
        m.texcoords = me.texcoords.ctypes.data_as(abi.C.POINTER(abi.C.c_float))
        m.indices = me.indices.ctypes.data_as(abi.C.POINTER(abi.C.c_uint32))
        m.n_vertices = me.n_vertices
        m.n_triangles = me.n_triangles
        m.transform = abi.f16(*cgmath.cols16(self.transform))
        m.inv_transform = abi.f16(*cgmath.cols16(self.inv_transform))
        m.material = fb.material(self.material) if self.material is not None else -1
        m.textures = abi.i5(*[fb.texture(t) if t is not None else -1 for t in self.textures])
        fb.add(abi.MI_OBJ_MESH, fb.meshes, m)


class FlatBuilder:
    """Collects PODs in Scene.objects order and materialises an mi_scene_desc."""

    def __init__(self):
        self.objects, self.spheres, self.triangles, self.planes = [], [], [], []
        self.volumes, self.meshes, self.materials, self.textures = [], [], [], []
        self._mat_ids, self._tex_ids, self._mesh_ids = {}, {}, {}
        self.boundary_objects = []
        self.keep = []

    def detached(self, obj):
        """Flatten `obj` into its typed array WITHOUT listing it in Scene.objects -> (kind, index)."""
        obj.flatten(self)
        o = self.objects.pop()
        return o.kind, o.index

    def add(self, kind, lst, pod):
        o = abi.mi_object()
        o.kind, o.index = kind, len(lst)
        lst.append(pod)
        self.objects.append(o)

    def material(self, mat: Material) -> int:
        key = id(mat)
        if key not in self._mat_ids:
            self._mat_ids[key] = len(self.materials)
            self.materials.append(mat.to_pod())
            self.keep.append(mat)
        return self._mat_ids[key]

    def texture(self, tex: Texture) -> int:
        key = id(tex)
        if key not in self._tex_ids:
            t = abi.mi_texture()
            t.width, t.height = tex.width, tex.height
            t.rgb = tex.img.ctypes.data_as(abi.C.POINTER(abi.C.c_uint8))
            self._tex_ids[key] = len(self.textures)
            self.textures.append(t)
            self.keep.append(tex)
        return self._tex_ids[key]

    def finish(self) -> "FlatScene":
        return FlatScene(self)


class FlatScene:
    """Owns the arrays an mi_scene_desc points into (borrowed by the library per call)."""

    def __init__(self, fb: FlatBuilder):
        self._fb = fb
        d = abi.mi_scene_desc()

        def arr(typ, items):
            a = (typ * max(1, len(items)))(*items)
            fb.keep.append(a)
            return a, len(items)

        d.objects, d.n_objects = arr(abi.mi_object, fb.objects)
        d.spheres, d.n_spheres = arr(abi.mi_sphere, fb.spheres)
        d.triangles, d.n_triangles = arr(abi.mi_triangle, fb.triangles)
        d.planes, d.n_planes = arr(abi.mi_plane, fb.planes)
        d.volumes, d.n_volumes = arr(abi.mi_volume, fb.volumes)
        d.meshes, d.n_meshes = arr(abi.mi_mesh, fb.meshes)
        d.materials, d.n_materials = arr(abi.mi_material, fb.materials)
        d.textures, d.n_textures = arr(abi.mi_texture, fb.textures)
        d.boundary_objects, d.n_boundary_objects = arr(abi.mi_object, fb.boundary_objects)
        self.desc = d
