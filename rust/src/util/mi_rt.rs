//! Rust binding of include/mi_rt.h for mbk6/CS397RayTracingSP22: what a maintainer of the reference adds as
//! `src/util/mi_rt.rs` (rust/README.md lists the other files of the patch set and the five lines of the reference they touch).
//! UNVERIFIED BY A COMPILER IN THIS CONTAINER: there is no cargo / rustc here.  What IS checked mechanically
//! (tests/test_abi.py): every `#[repr(C)]` struct below against the C header field by field (order, type, count), every
//! `extern "C"` function against the header's prototypes (name, argument count and types), the numeric constants, and the
//! bracket structure of every file of the patch set.
#![allow(non_camel_case_types, dead_code)]
use std::collections::HashMap;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Clone, Copy)] pub struct mi_material { pub kind: i32, pub albedo: [f32; 3], pub emission: [f32; 3], pub roughness: f32, pub metallic: f32, pub idx_of_refraction: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct mi_object { pub kind: i32, pub index: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct mi_sphere { pub center: [f32; 3], pub radius: f32, pub material: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct mi_triangle { pub a: [f32; 3], pub b: [f32; 3], pub c: [f32; 3], pub material: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct mi_plane { pub point: [f32; 3], pub normal: [f32; 3], pub material: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct mi_volume { pub boundary_center: [f32; 3], pub boundary_radius: f32, pub density: f32, pub phase_material: i32,
    pub boundary_kind: i32, pub boundary_index: i32, pub boundary_count: i32 }   // 0 = the inline sphere; else Triangle / Plane / StaticMesh / nested Scene
#[repr(C)] #[derive(Clone, Copy)] pub struct mi_texture { pub width: i32, pub height: i32, pub rgb: *const u8 }
#[repr(C)] #[derive(Clone, Copy)] pub struct mi_mesh {
    pub positions: *const f32, pub normals: *const f32, pub texcoords: *const f32, pub indices: *const u32,
    pub n_vertices: i32, pub n_triangles: i32, pub transform: [f32; 16], pub inv_transform: [f32; 16],
    pub material: i32, pub textures: [i32; 5],
}
#[repr(C)] pub struct mi_scene_desc {
    pub objects: *const mi_object, pub n_objects: i32, pub spheres: *const mi_sphere, pub n_spheres: i32,
    pub triangles: *const mi_triangle, pub n_triangles: i32, pub planes: *const mi_plane, pub n_planes: i32,
    pub volumes: *const mi_volume, pub n_volumes: i32, pub meshes: *const mi_mesh, pub n_meshes: i32,
    pub materials: *const mi_material, pub n_materials: i32, pub textures: *const mi_texture, pub n_textures: i32,
    pub boundary_objects: *const mi_object, pub n_boundary_objects: i32,   // entries of nested Scenes used as ConvexVolume boundaries
    pub point_light_pos: [f32; 3], pub ambient: [f32; 3],   // Scene.point_light_pos / ambient (tracing.rs:216-217), Phong only
}
#[repr(C)] pub struct mi_camera_desc {
    pub eyepoint: [f32; 3], pub view_dir: [f32; 3], pub up: [f32; 3], pub projection_mode: i32, pub shading_mode: i32,
    pub path_depth: u32, pub path_samples: u32, pub screen_width: u32, pub screen_height: u32,
    pub focal_length: f32, pub focus_dist: f32, pub lens_radius: f32, pub aa_sample_count: u32, pub max_trace_dist: f32, pub gamma: f32,
}
#[repr(C)] #[derive(Default)] pub struct mi_render_opts { pub seed: u32, pub rank: i32, pub world: i32, pub variant: i32, pub want_signature: i32, pub flags: u32, pub max_state_bytes: u64 }
#[repr(C)] #[derive(Default)] pub struct mi_stats { pub samples: u64, pub pixels: u64, pub tiles: u32, pub tiles_padded: u32, pub kernel_ms: f32, pub total_ms: f32, pub scene_bytes: u32, pub scene_in_lds: u32 }
pub enum mi_ctx {}
pub enum mi_multi {}
pub const MI_OPT_NO_TILE_MASKS: u32 = 1; pub const MI_OPT_REFERENCE_WALK: u32 = 2; pub const MI_OPT_TWO_STAGE: u32 = 4; pub const MI_OPT_NO_LIST_TREE: u32 = 8;
pub const MI_RT_ABI_VERSION: c_int = 5;
// mi_material_kind / mi_object_kind / projection and shading modes (include/mi_rt.h)
pub const MI_MAT_LAMBERTIAN: i32 = 0; pub const MI_MAT_METAL: i32 = 1; pub const MI_MAT_DIELECTRIC: i32 = 2;
pub const MI_MAT_PARAMETERIZED: i32 = 3; pub const MI_MAT_ISOTROPIC: i32 = 4;
pub const MI_OBJ_SPHERE: i32 = 0; pub const MI_OBJ_TRIANGLE: i32 = 1; pub const MI_OBJ_PLANE: i32 = 2;
pub const MI_OBJ_VOLUME: i32 = 3; pub const MI_OBJ_MESH: i32 = 4; pub const MI_OBJ_SCENE: i32 = 5;
pub const MI_PROJ_ORTHOGRAPHIC: i32 = 0; pub const MI_PROJ_PERSPECTIVE: i32 = 1;
pub const MI_SHADE_PHONG: i32 = 0; pub const MI_SHADE_PATHTRACE: i32 = 1;

#[link(name = "mi_rt")]
extern "C" {
    // every entry point of include/mi_rt.h (ABI 5), in header order
    pub fn mi_ctx_create(device: c_int, out: *mut *mut mi_ctx) -> c_int;
    pub fn mi_ctx_destroy(ctx: *mut mi_ctx);
    pub fn mi_scene_upload(ctx: *mut mi_ctx, scene: *const mi_scene_desc) -> c_int;
    pub fn mi_render(ctx: *mut mi_ctx, cam: *const mi_camera_desc, opts: *const mi_render_opts,
                     out_rgb_f32: *mut f32, out_rgb_u8: *mut u8, out_sig: *mut u32, stats: *mut mi_stats) -> c_int;
    pub fn mi_compact_size(cam: *const mi_camera_desc, world: i32, tiles_total: *mut u32, tiles_padded: *mut u32) -> c_int;
    // device-pointer building blocks (one process per GPU); pointers are device addresses, `stream` a hipStream_t
    pub fn mi_render_tiles_device(ctx: *mut mi_ctx, cam: *const mi_camera_desc, opts: *const mi_render_opts,
                                  d_compact_f32: *mut c_void, d_sig_u32: *mut c_void, stream: *mut c_void, stats: *mut mi_stats) -> c_int;
    pub fn mi_render_samples_device(ctx: *mut mi_ctx, cam: *const mi_camera_desc, opts: *const mi_render_opts,
                                    sample_begin: u32, sample_end: u32, d_accum_f32x4: *mut c_void,
                                    d_compact_f32: *mut c_void, d_sig_u32: *mut c_void, stream: *mut c_void, stats: *mut mi_stats) -> c_int;
    pub fn mi_unpermute_device(ctx: *mut mi_ctx, cam: *const mi_camera_desc, world: i32,
                               d_gathered_f32: *const c_void, d_image_f32: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn mi_tonemap_device(ctx: *mut mi_ctx, cam: *const mi_camera_desc, d_image_f32: *const c_void, d_image_u8: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn mi_last_kernel_ms(ctx: *mut mi_ctx, ms: *mut f32) -> c_int;
    pub fn mi_reserve(ctx: *mut mi_ctx, cam: *const mi_camera_desc, world: i32, max_state_bytes: u64) -> c_int;
    pub fn mi_last_pipeline_ms(ctx: *mut mi_ctx, out8: *mut f32) -> c_int;
    pub fn mi_last_pipeline_counts(ctx: *mut mi_ctx, out8: *mut u64) -> c_int;
    pub fn mi_last_diag(ctx: *mut mi_ctx, out16: *mut u64) -> c_int;
    pub fn mi_selftest(ctx: *mut mi_ctx, out4: *mut u64) -> c_int;
    // N GPUs of one node behind one blocking call (RCCL fan-in inside the library)
    pub fn mi_multi_create(n_devices: c_int, devices: *const c_int, out: *mut *mut mi_multi) -> c_int;
    pub fn mi_multi_create_loopback(n_contexts: c_int, device: c_int, out: *mut *mut mi_multi) -> c_int;   // test transport: N ranks on one device
    pub fn mi_multi_destroy(m: *mut mi_multi);
    pub fn mi_multi_device_count(m: *const mi_multi) -> c_int;
    pub fn mi_multi_context(m: *const mi_multi, rank: c_int) -> *mut mi_ctx;
    pub fn mi_multi_scene_upload(m: *mut mi_multi, scene: *const mi_scene_desc) -> c_int;
    pub fn mi_multi_reserve(m: *mut mi_multi, cam: *const mi_camera_desc, max_state_bytes: u64) -> c_int;
    pub fn mi_multi_render(m: *mut mi_multi, cam: *const mi_camera_desc, opts: *const mi_render_opts,
                           out_rgb_f32: *mut f32, out_rgb_u8: *mut u8, out_sig: *mut u32, stats: *mut mi_stats) -> c_int;
    pub fn mi_last_error() -> *const c_char;
    pub fn mi_abi_version() -> c_int;           // 5: assert at start-up that header and library agree
}


/// The additive half of `trait Intersectable` (tracing.rs:42-47): `pub trait Intersectable: mi_rt::FlattenObject`.
/// A supertrait rather than a new method, so that the implementations live in their own `impl` blocks (the
/// `*_flatten.rs` fragments) and no existing `impl Intersectable for ..` block of the reference is edited.
/// The default body is what the helper types get that never appear in `Scene.objects` (AABB, BVHNode, IndexedTriangle).
pub trait FlattenObject {
    fn flatten(&self, out: &mut SceneBuilder) { out.unsupported(std::any::type_name::<Self>()); }
}
/// The additive half of `trait Material` (materials.rs:12-15): `pub trait Material: mi_rt::FlattenMaterial`.
pub trait FlattenMaterial {
    fn flatten(&self) -> mi_material;
}

/// Collects the PODs of a scene in `Scene.objects` order (that order decides ties, tracing.rs:335, and the order of the
/// RNG draws of ConvexVolume::intersect_ray, geometry.rs:517).  Owns what the PODs point into where the reference does not:
/// the RGB8 bytes of the textures.  Mesh arrays are borrowed from the `Arc<Mesh>` of the StaticMesh, which outlives the call.
#[derive(Default)]
pub struct SceneBuilder {
    pub objects: Vec<mi_object>, pub spheres: Vec<mi_sphere>, pub triangles: Vec<mi_triangle>, pub planes: Vec<mi_plane>,
    pub volumes: Vec<mi_volume>, pub meshes: Vec<mi_mesh>, pub materials: Vec<mi_material>, pub textures: Vec<mi_texture>,
    pub boundary_objects: Vec<mi_object>,
    /// first reason why this scene cannot run on the GPU path (None = it can)
    pub unsupported: Option<String>,
    tex_bytes: Vec<Vec<u8>>,                   // backing store of `textures[i].rgb` (a Vec's heap block does not move when the outer Vec grows)
    material_ids: HashMap<usize, i32>,         // Arc<dyn Material> data pointer -> index: a shared Arc is one mi_material
    mesh_ids: HashMap<usize, i32>,             // &StaticMesh address -> index: the same Arc<StaticMesh> listed twice is ONE mi_mesh
    detached: Vec<Vec<mi_object>>,             // open ConvexVolume boundaries: entries pushed meanwhile are NOT Scene.objects entries
}
impl SceneBuilder {
    pub fn unsupported(&mut self, what: &str) {
        if self.unsupported.is_none() { self.unsupported = Some(what.to_string()); }
    }
    /// One entry of Scene.objects — or, inside a ConvexVolume boundary, one entry of that boundary.
    pub fn push_object(&mut self, kind: i32, index: i32) {
        let o = mi_object { kind: kind, index: index };
        match self.detached.last_mut() { Some(list) => list.push(o), None => self.objects.push(o) }
    }
    pub fn in_boundary(&self) -> bool { !self.detached.is_empty() }
    pub fn begin_boundary(&mut self) { self.detached.push(Vec::new()); }
    pub fn end_boundary(&mut self) -> Vec<mi_object> { self.detached.pop().unwrap_or_default() }
    /// Index of `m` in `materials`; an Arc seen before (same data pointer) is not pushed again.
    pub fn material(&mut self, m: &std::sync::Arc<dyn super::materials::Material + Send + Sync>) -> i32 {
        let key = std::sync::Arc::as_ptr(m) as *const () as usize;
        if let Some(&i) = self.material_ids.get(&key) { return i; }
        let i = self.materials.len() as i32;
        self.materials.push(m.flatten());
        self.material_ids.insert(key, i);
        i
    }
    /// Index of a texture given as tightly packed RGB8 rows, top row first (texture.rs:30 `get_pixel(x, y).to_rgb()`).
    pub fn texture_rgb8(&mut self, width: u32, height: u32, rgb: Vec<u8>) -> i32 {
        assert_eq!(rgb.len(), width as usize * height as usize * 3);
        let i = self.textures.len() as i32;
        self.tex_bytes.push(rgb);
        let p = self.tex_bytes.last().unwrap().as_ptr();
        self.textures.push(mi_texture { width: width as i32, height: height as i32, rgb: p });
        i
    }
    pub fn mesh_index(&self, key: usize) -> Option<i32> { self.mesh_ids.get(&key).copied() }
    pub fn remember_mesh(&mut self, key: usize, index: i32) { self.mesh_ids.insert(key, index); }

    /// The POD view; valid while `self` (and the scene it was flattened from) is alive and unchanged.
    pub fn desc(&self) -> mi_scene_desc {
        mi_scene_desc {
            objects: self.objects.as_ptr(), n_objects: self.objects.len() as i32,
            spheres: self.spheres.as_ptr(), n_spheres: self.spheres.len() as i32,
            triangles: self.triangles.as_ptr(), n_triangles: self.triangles.len() as i32,
            planes: self.planes.as_ptr(), n_planes: self.planes.len() as i32,
            volumes: self.volumes.as_ptr(), n_volumes: self.volumes.len() as i32,
            meshes: self.meshes.as_ptr(), n_meshes: self.meshes.len() as i32,
            materials: self.materials.as_ptr(), n_materials: self.materials.len() as i32,
            textures: self.textures.as_ptr(), n_textures: self.textures.len() as i32,
            boundary_objects: self.boundary_objects.as_ptr(), n_boundary_objects: self.boundary_objects.len() as i32,
            point_light_pos: [0.0, 1.0, 5.0], ambient: [0.1, 0.1, 0.1],   // the caller overwrites both from Scene (tracing.rs:216-217)
        }
    }
}

/// `mi_last_error()` of this thread as a String.
pub fn last_error() -> String {
    unsafe {
        let p = mi_last_error();
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
