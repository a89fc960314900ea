// geometry_flatten.rs — pasted into src/util/geometry.rs by `include!("geometry_flatten.rs");` (last line of that file:
// StaticMesh's fields are private, geometry.rs:127-134, so its impl has to live in that module).
// The `Intersectable` implementors as PODs of include/mi_rt.h, in the order and with the sharing the reference's
// `Scene.objects: Arc<Vec<Arc<dyn Intersectable>>>` (tracing.rs:215) has.
// Needs: `pub trait Intersectable: super::mi_rt::FlattenObject` (tracing.rs:42).  UNVERIFIED by a compiler (see mi_rt.rs).

use super::mi_rt::{self, FlattenObject, SceneBuilder};

// Helper types that implement Intersectable but are never entries of Scene.objects: the default body (unsupported).
impl FlattenObject for AABB {}                           // geometry.rs:50
impl FlattenObject for BVHNode {}                        // geometry.rs:93  (the BVH is rebuilt inside mi_scene_upload with this topology)
impl FlattenObject for IndexedTriangle {}                // geometry.rs:330

impl FlattenObject for Sphere {                          // geometry.rs:389-393
    fn flatten(&self, out: &mut SceneBuilder) {
        let m = out.material(&self.material);
        let index = out.spheres.len() as i32;
        out.spheres.push(mi_rt::mi_sphere { center: self.center.into(), radius: self.radius, material: m });
        out.push_object(mi_rt::MI_OBJ_SPHERE, index);
    }
}

impl FlattenObject for Triangle {                        // geometry.rs:424-429
    fn flatten(&self, out: &mut SceneBuilder) {
        let m = out.material(&self.material);
        let index = out.triangles.len() as i32;
        out.triangles.push(mi_rt::mi_triangle { a: self.a.into(), b: self.b.into(), c: self.c.into(), material: m });
        out.push_object(mi_rt::MI_OBJ_TRIANGLE, index);
    }
}

impl FlattenObject for Plane {                           // geometry.rs:468-472
    fn flatten(&self, out: &mut SceneBuilder) {
        let m = out.material(&self.material);
        let index = out.planes.len() as i32;
        out.planes.push(mi_rt::mi_plane { point: self.point.into(), normal: self.normal.into(), material: m });
        out.push_object(mi_rt::MI_OBJ_PLANE, index);
    }
}

impl FlattenObject for ConvexVolume {                    // geometry.rs:495-500
    // `boundary` is an Arc<dyn Intersectable>; its intersect_ray is called twice per ray (geometry.rs:505,508) and its own
    // material is ignored (tracing.rs:503 "arbitrary").  It is flattened with the entries it pushes DIVERTED from Scene.objects:
    //   one Sphere (every use in the reference, tracing.rs:499-516) -> inline centre + radius, boundary_kind 0;
    //   one Triangle / Plane / StaticMesh                           -> boundary_kind = its kind, boundary_index;
    //   anything that pushed several entries (a nested Scene, tracing.rs:326) -> a run of boundary_objects, MI_OBJ_SCENE.
    // A ConvexVolume met while a boundary is open is what the GPU path does not do (MI_ERR_UNSUPPORTED at the ABI as well).
    fn flatten(&self, out: &mut SceneBuilder) {
        if out.in_boundary() { out.unsupported("a ConvexVolume inside a ConvexVolume boundary"); return; }
        let spheres_before = out.spheres.len();
        out.begin_boundary();
        self.boundary.flatten(out);
        let entries = out.end_boundary();
        let mut v = mi_rt::mi_volume { boundary_center: [0.0; 3], boundary_radius: 0.0, density: self.density, phase_material: 0,
                                       boundary_kind: mi_rt::MI_OBJ_SPHERE, boundary_index: 0, boundary_count: 0 };
        if entries.len() == 1 && entries[0].kind == mi_rt::MI_OBJ_SPHERE && out.spheres.len() == spheres_before + 1 {
            let s = out.spheres.pop().unwrap();          // the boundary sphere is not an object of the scene: it travels inline
            v.boundary_center = s.center; v.boundary_radius = s.radius;
        } else if entries.len() == 1 {
            v.boundary_kind = entries[0].kind; v.boundary_index = entries[0].index;
        } else {
            v.boundary_kind = mi_rt::MI_OBJ_SCENE;
            v.boundary_index = out.boundary_objects.len() as i32; v.boundary_count = entries.len() as i32;
            out.boundary_objects.extend(entries);
        }
        v.phase_material = out.material(&self.phase_function);
        let index = out.volumes.len() as i32;
        out.volumes.push(v);
        out.push_object(mi_rt::MI_OBJ_VOLUME, index);
    }
}

impl FlattenObject for StaticMesh {                      // geometry.rs:127-134
    // tobj's single-index Mesh (geometry.rs:140-148) goes over by pointer: positions / normals / texcoords / indices stay in
    // the Arc<Mesh> this StaticMesh holds.  The same Arc<StaticMesh> listed twice in Scene.objects (address of `self`) is ONE
    // mi_mesh and two entries.  normals and texcoords are required (geometry.rs:350,355 index them on every candidate hit).
    fn flatten(&self, out: &mut SceneBuilder) {
        let key = self as *const StaticMesh as usize;
        if let Some(index) = out.mesh_index(key) { out.push_object(mi_rt::MI_OBJ_MESH, index); return; }
        let me = &*self.mesh;
        let n_vertices = me.positions.len() / 3;
        if me.normals.len() != me.positions.len() || me.texcoords.len() * 3 != me.positions.len() * 2 || me.indices.len() % 3 != 0 {
            out.unsupported("StaticMesh without per-vertex normals and texcoords (geometry.rs:350,355 would panic)");
            return;
        }
        let material = match &self.material { Some(m) => out.material(m), None => -1 };      // geometry.rs:255: None = from textures
        let mut textures = [-1i32; 5];                  // 0 albedo 1 emission 2 metallic 3 roughness 4 normal (geometry.rs:130)
        for k in 0..5 {
            if let Some(t) = &self.textures[k] { textures[k] = t.flatten(out); }
        }
        let transform: &[f32; 16] = self.transform.as_ref();            // cgmath Matrix4 is column-major, as mi_mesh wants
        let inv_transform: &[f32; 16] = self.inv_transform.as_ref();    // geometry.rs:168
        let index = out.meshes.len() as i32;
        out.meshes.push(mi_rt::mi_mesh {
            positions: me.positions.as_ptr(), normals: me.normals.as_ptr(), texcoords: me.texcoords.as_ptr(), indices: me.indices.as_ptr(),
            n_vertices: n_vertices as i32, n_triangles: (me.indices.len() / 3) as i32,
            transform: *transform, inv_transform: *inv_transform, material: material, textures: textures,
        });
        out.remember_mesh(key, index);
        out.push_object(mi_rt::MI_OBJ_MESH, index);
    }
}
