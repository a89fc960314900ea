//! Source-only Rust binding of include/mi_rt.h for mbk6/CS397RayTracingSP22.
//! UNVERIFIED IN THIS CONTAINER: there is no cargo/rustc here, so this file has never been
//! compiled.  It is what a maintainer of the reference would add as `src/util/mi_rt.rs`
//! (plus `mod mi_rt;` in src/util.rs and `println!("cargo:rustc-link-lib=mi_rt")` in build.rs).
#![allow(non_camel_case_types, dead_code)]
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
pub const MI_OPT_NO_TILE_MASKS: u32 = 1; pub const MI_OPT_REFERENCE_WALK: u32 = 2; pub const MI_OPT_TWO_STAGE: u32 = 4;

#[link(name = "mi_rt")]
extern "C" {
    // every entry point of include/mi_rt.h (ABI 4), in header order
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
    pub fn mi_multi_destroy(m: *mut mi_multi);
    pub fn mi_multi_device_count(m: *const mi_multi) -> c_int;
    pub fn mi_multi_context(m: *const mi_multi, rank: c_int) -> *mut mi_ctx;
    pub fn mi_multi_scene_upload(m: *mut mi_multi, scene: *const mi_scene_desc) -> c_int;
    pub fn mi_multi_reserve(m: *mut mi_multi, cam: *const mi_camera_desc, max_state_bytes: u64) -> c_int;
    pub fn mi_multi_render(m: *mut mi_multi, cam: *const mi_camera_desc, opts: *const mi_render_opts,
                           out_rgb_f32: *mut f32, out_rgb_u8: *mut u8, out_sig: *mut u32, stats: *mut mi_stats) -> c_int;
    pub fn mi_last_error() -> *const c_char;
    pub fn mi_abi_version() -> c_int;           // 4: assert at start-up that header and library agree
}

/// Collected in `Scene.objects` order by the additive trait method
/// `fn flatten(&self, out: &mut SceneBuilder)` on `Intersectable` (tracing.rs:42-47) and
/// `fn flatten(&self) -> mi_material` on `Material` (materials.rs:12-15).
#[derive(Default)]
pub struct SceneBuilder {
    pub objects: Vec<mi_object>, pub spheres: Vec<mi_sphere>, pub triangles: Vec<mi_triangle>, pub planes: Vec<mi_plane>,
    pub volumes: Vec<mi_volume>, pub meshes: Vec<mi_mesh>, pub materials: Vec<mi_material>, pub textures: Vec<mi_texture>,
    pub boundary_objects: Vec<mi_object>,
}
impl SceneBuilder {
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
            point_light_pos: [0.0, 1.0, 5.0], ambient: [0.1, 0.1, 0.1],   // caller overwrites from Scene
        }
    }
}
pub fn _unused(_: *mut c_void) {}
