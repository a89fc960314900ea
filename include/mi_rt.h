/*
 * mi_rt.h — C ABI of the MI355X path-tracing hot path (libmi_rt.so).
 *
 * This header is the drop-in boundary for ONE path of mbk6/CS397RayTracingSP22:
 *     Scene::render_to_image()            src/util/tracing.rs:221-263
 *       -> Camera::generate_rays()        src/util/tracing.rs:159-209
 *       -> Scene::shade_ray()             src/util/tracing.rs:300-324
 *       -> Scene::intersect_ray()         src/util/tracing.rs:326-346
 *            -> {Sphere,Triangle,Plane,ConvexVolume,StaticMesh}::intersect_ray
 *                                         src/util/geometry.rs:300-321, 394-413, 430-450, 473-489, 501-526
 *            -> BVHNode / AABB / IndexedTriangle   src/util/geometry.rs:50-79, 93-119, 330-366
 *       -> Material::scatter()/emission() src/util/materials.rs:33-48, 56-71, 77-104, 113-149, 158-166
 *       -> Texture::sample()              src/util/texture.rs:26-32
 *
 * The reference has no FFI of its own (it is one Rust crate; the path is reached by
 * ordinary calls through `dyn Intersectable` / `dyn Material`).  What crosses this
 * boundary is therefore the *flattened* form of the reference's own structs: every
 * struct below mirrors one reference struct field for field, and `mi_scene_desc.objects`
 * keeps the order of `Scene.objects` (tracing.rs:215) because that order decides ties
 * (tracing.rs:335, strict `<`: the first object wins) and the order of RNG draws made by
 * ConvexVolume::intersect_ray (geometry.rs:517).
 *
 * Plain C: pointers, sizes, PODs.  No torch / HIP types in any signature; device
 * pointers and streams travel as `void*`.  Nothing unwinds across this boundary:
 * every entry point returns an mi_status and records a message for mi_last_error().
 *
 * Ownership: every input array is borrowed for the duration of the call only
 * (mi_scene_upload copies to the device).  Output buffers are allocated by the caller.
 * A mi_ctx owns its device memory; it is not re-entrant, distinct contexts may be
 * used from distinct threads.
 *
 * All arithmetic on the path is f32 (Vec3 = Vector3<f32>, tracing.rs:22).
 */
#ifndef MI_RT_H
#define MI_RT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_RT_ABI_VERSION 5

/* ---- status codes (reference: panics via assert!/expect/unwrap, geometry.rs:149-151) ---- */
typedef enum mi_status {
    MI_OK               =  0,
    MI_ERR_INVALID      = -1,  /* NULL pointer, bad index, bad size                      */
    MI_ERR_UNSUPPORTED  = -2,  /* reference feature outside the accelerated path         */
    MI_ERR_NO_DEVICE    = -3,  /* no gfx950 device / HIP runtime failure at create       */
    MI_ERR_HIP          = -4,  /* a HIP call failed (message holds hipGetErrorString)    */
    MI_ERR_OOM          = -5,
    MI_ERR_NO_SCENE     = -6   /* render called before a scene was uploaded              */
} mi_status;

/* ---- materials: `trait Material` implementors, materials.rs:20,51,74,107,152 ---- */
typedef enum mi_material_kind {
    MI_MAT_LAMBERTIAN    = 0,  /* Lambertian{albedo, emission}                      :20  */
    MI_MAT_METAL         = 1,  /* Metal{albedo, emission, roughness}                :51  */
    MI_MAT_DIELECTRIC    = 2,  /* Dielectric{idx_of_refraction}                     :74  */
    MI_MAT_PARAMETERIZED = 3,  /* ParameterizedMaterial{albedo,emission,rough,metal}:107 */
    MI_MAT_ISOTROPIC     = 4   /* Isotropic{albedo, emission}                       :152 */
} mi_material_kind;

typedef struct mi_material {
    int32_t kind;              /* mi_material_kind */
    float   albedo[3];
    float   emission[3];
    float   roughness;
    float   metallic;
    float   idx_of_refraction;
} mi_material;                 /* 40 bytes */

/* ---- primitives: `trait Intersectable` implementors, geometry.rs:389,424,468,495,127 ---- */
typedef enum mi_object_kind {
    MI_OBJ_SPHERE   = 0,       /* Sphere{center, radius, material}          geometry.rs:389 */
    MI_OBJ_TRIANGLE = 1,       /* Triangle{a, b, c, material}               geometry.rs:424 */
    MI_OBJ_PLANE    = 2,       /* Plane{point, normal, material}            geometry.rs:468 */
    MI_OBJ_VOLUME   = 3,       /* ConvexVolume{boundary, phase_function, density}      :495 */
    MI_OBJ_MESH     = 4,       /* StaticMesh                                geometry.rs:127 */
    MI_OBJ_SCENE    = 5        /* a nested Scene (`impl Intersectable for Scene`, tracing.rs:326) — only as the boundary of a ConvexVolume */
} mi_object_kind;

/* One entry of Scene.objects (tracing.rs:215), in the reference's order. */
typedef struct mi_object {
    int32_t kind;              /* mi_object_kind                         */
    int32_t index;             /* index into the typed array of that kind */
} mi_object;

typedef struct mi_sphere   { float center[3]; float radius; int32_t material; } mi_sphere;
typedef struct mi_triangle { float a[3]; float b[3]; float c[3]; int32_t material; } mi_triangle;
typedef struct mi_plane    { float point[3]; float normal[3]; int32_t material; } mi_plane;

/* ConvexVolume (geometry.rs:495-500).  `boundary` is `Arc<dyn Intersectable>` in the reference and its intersect_ray is called
 * twice per ray (geometry.rs:505,508: entry with t in [f32::MIN, f32::MAX], exit from t_entr + 1e-4).  boundary_kind says what it is:
 *   MI_OBJ_SPHERE (0, what every use in the reference is, tracing.rs:499-516): the sphere given INLINE by boundary_center / _radius;
 *   MI_OBJ_TRIANGLE, MI_OBJ_PLANE, MI_OBJ_MESH: entry boundary_index of the scene's typed array of that kind;
 *   MI_OBJ_SCENE: a nested Scene (its closest hit, first entry wins ties, tracing.rs:330-344) = the boundary_count entries of
 *                 mi_scene_desc.boundary_objects starting at boundary_index, each a Sphere / Triangle / Plane / StaticMesh.
 * The boundary object need not be listed in Scene.objects; its own material is ignored by the reference ("arbitrary",
 * tracing.rs:503).  A ConvexVolume or a Scene INSIDE a boundary is MI_ERR_UNSUPPORTED. */
typedef struct mi_volume {
    float   boundary_center[3];
    float   boundary_radius;
    float   density;
    int32_t phase_material;    /* index of the phase-function material (Isotropic) */
    int32_t boundary_kind;     /* mi_object_kind, 0 = the inline sphere */
    int32_t boundary_index;
    int32_t boundary_count;    /* MI_OBJ_SCENE only */
} mi_volume;

/* Texture (texture.rs:12-14) after `get_pixel(..).to_rgb()`: tightly packed RGB8,
 * row 0 = top row of the image file.  Decoding image files is load-time work outside
 * this path (texture.rs:16-25). */
typedef struct mi_texture {
    int32_t        width;
    int32_t        height;
    const uint8_t* rgb;        /* width*height*3 bytes */
} mi_texture;

/* StaticMesh (geometry.rs:127-134).  Geometry is tobj's single-index `Mesh`
 * (geometry.rs:140-148): one index per corner into positions/normals/texcoords.
 * normals and texcoords are required: the reference indexes them on every candidate
 * hit (geometry.rs:350,355) and would panic without them. */
typedef struct mi_mesh {
    const float*    positions;      /* n_vertices*3 */
    const float*    normals;        /* n_vertices*3 */
    const float*    texcoords;      /* n_vertices*2 */
    const uint32_t* indices;        /* n_triangles*3 */
    int32_t         n_vertices;
    int32_t         n_triangles;
    float           transform[16];      /* cgmath Matrix4, column-major     geometry.rs:132 */
    float           inv_transform[16];  /* transform.inverse_transform()    geometry.rs:168 */
    int32_t         material;           /* index, or -1 = material from textures (:255)     */
    int32_t         textures[5];        /* 0 albedo 1 emission 2 metallic 3 roughness 4 normal; -1 = None (:130) */
} mi_mesh;

typedef struct mi_scene_desc {
    const mi_object*   objects;    int32_t n_objects;
    const mi_sphere*   spheres;    int32_t n_spheres;
    const mi_triangle* triangles;  int32_t n_triangles;
    const mi_plane*    planes;     int32_t n_planes;
    const mi_volume*   volumes;    int32_t n_volumes;
    const mi_mesh*     meshes;     int32_t n_meshes;
    const mi_material* materials;  int32_t n_materials;
    const mi_texture*  textures;   int32_t n_textures;
    /* entries of the nested Scenes that serve as ConvexVolume boundaries (mi_volume.boundary_kind == MI_OBJ_SCENE); may be NULL / 0 */
    const mi_object*   boundary_objects;  int32_t n_boundary_objects;
    /* Scene.point_light_pos / Scene.ambient (tracing.rs:216-217): read by ShadingMode::Phong only */
    float              point_light_pos[3];
    float              ambient[3];
} mi_scene_desc;

/* ---- Camera (tracing.rs:138-155), field for field ---- */
#define MI_PROJ_ORTHOGRAPHIC 0    /* CameraProjectionMode::Orthographic (tracing.rs:196,200)           */
#define MI_PROJ_PERSPECTIVE  1
#define MI_SHADE_PHONG       0    /* ShadingMode::Phong, the debug shader (tracing.rs:277-297); own kernel, ignores `variant` */
#define MI_SHADE_PATHTRACE   1

typedef struct mi_camera_desc {
    float    eyepoint[3];
    float    view_dir[3];
    float    up[3];
    int32_t  projection_mode;
    int32_t  shading_mode;
    uint32_t path_depth;
    uint32_t path_samples;      /* 1 in every configuration (tracing.rs:370); != 1 runs MI_VARIANT_RECURSIVE */
    uint32_t screen_width;
    uint32_t screen_height;
    float    focal_length;
    float    focus_dist;
    float    lens_radius;
    uint32_t aa_sample_count;   /* perfect square (tracing.rs:152) */
    float    max_trace_dist;
    float    gamma;
} mi_camera_desc;

/* ---- render options (new: the reference RNG is unseeded thread_rng) ---- */
#define MI_TILE 32              /* image tiles are MI_TILE x MI_TILE pixels */

typedef struct mi_render_opts {
    uint32_t seed;              /* RNG stream key: (seed, y*W+x, sample)                 */
    int32_t  rank;              /* this process renders tiles t with t % world == rank   */
    int32_t  world;             /* number of ranks sharing the image (>=1)               */
    int32_t  variant;           /* 0 = default kernel; see mi_variant                    */
    int32_t  want_signature;    /* 1 = also produce per-pixel path signatures (diagnostic) */
    uint32_t flags;             /* MI_OPT_* bits, 0 = defaults                           */
    uint64_t max_state_bytes;   /* wavefront pipeline: upper bound on the HBM it may hold for path state and
                                 * sample slots (0 = what renders the frame in ONE batch, at most 60 % of the free HBM: a
                                 * whole 1080p / 256 spp frame is 110 GB).  A smaller budget means more, smaller sample
                                 * batches: same image, bit for bit, lower throughput (cfg2, round 4: one batch 72.7 ms;
                                 * 64 / 32 / 16 / 8 / 4 GB: +1 / +8 / +13 / +24 / +43 %: DESIGN.md section 4).
                                 * The smallest batch is one sample of every pixel of the rank (about 210 B per pixel, 280 B
                                 * with a two-stage mesh): a non-zero budget below that is MI_ERR_INVALID, never silently exceeded */
} mi_render_opts;               /* 32 bytes */

#define MI_OPT_NO_TILE_MASKS   1u   /* camera rays test every Scene.objects entry (no per-tile frustum masks): same image */
#define MI_OPT_REFERENCE_WALK  2u   /* meshes: walk the reference's own BVH (geometry.rs:94-119) node by node even where the
                                     * exact two-stage traversal would be used: same image, slower on large meshes */
#define MI_OPT_TWO_STAGE       4u   /* meshes: use the two-stage traversal for every mesh, whatever its size: same image */
#define MI_OPT_NO_LIST_TREE    8u   /* long lists: test every Triangle of Scene.objects one by one instead of walking the top-level tree the scene
                                     * compiler builds over them (>= 32 small triangles): same image */

typedef enum mi_variant {
    MI_VARIANT_DEFAULT    = 0,  /* library picks (currently MI_VARIANT_WAVEFRONT)                  */
    MI_VARIANT_SIMPLE     = 1,  /* one segment per loop trip, mesh traversal in line (structural cross-check) */
    /* 2 was MI_VARIANT_PARKED: removed in ABI 3 (documented slower)            */
    MI_VARIANT_VOTED      = 3,  /* per-lane state machine; every BVH node step is a __ballot-voted phase */
    MI_VARIANT_VOTED_DIAG = 4,  /* VOTED + per-phase trip / active-lane counters (never timed)      */
    /* 5, 6 were MI_VARIANT_POOLED(_DIAG): removed in ABI 3                                          */
    MI_VARIANT_WAVEFRONT  = 7,  /* path state streamed through HBM, one kernel per phase (K1w); synchronises the stream */
    MI_VARIANT_RECURSIVE  = 8   /* shade_ray as written (tracing.rs:300-324), recursion on a per-lane stack: the only variant for
                                 * path_samples != 1 (chosen automatically); slow, bit-identical f32 image to the CPU restatement */
} mi_variant;

typedef struct mi_stats {
    uint64_t samples;           /* camera rays (paths) traced by this call                */
    uint64_t pixels;            /* pixels owned by this rank                              */
    uint32_t tiles;             /* tiles owned by this rank                               */
    uint32_t tiles_padded;      /* ceil(total_tiles/world): slots in the compact buffer   */
    float    kernel_ms;         /* path-tracing kernel, HIP events on the launch stream   */
    float    total_ms;          /* whole call incl. un-permute / tone-map / copies        */
    uint32_t scene_bytes;       /* bytes of scene data resident on device                 */
    uint32_t scene_in_lds;      /* 1 if the mesh BVH was staged into LDS                  */
} mi_stats;

typedef struct mi_ctx mi_ctx;

/* Create a context on HIP device `device` (one process per GPU: pass LOCAL_RANK). */
int  mi_ctx_create(int device, mi_ctx** out);
void mi_ctx_destroy(mi_ctx* ctx);

/* Flatten + upload a scene: replaces the construction of `Scene.objects`
 * (tracing.rs:374-540) and StaticMesh::build_bvh (geometry.rs:175-217; the BVH is
 * rebuilt here with the reference's topology). */
int  mi_scene_upload(mi_ctx* ctx, const mi_scene_desc* scene);

/* Replaces Scene::render_to_image (tracing.rs:221-263) for a whole image on one GPU.
 * out_rgb_f32: W*H*3 linear per-pixel means BEFORE saturation/gamma (parity surface), may be NULL.
 * out_rgb_u8 : W*H*3, the RgbImage byte layout of tracing.rs:226,254-256, may be NULL.
 * out_sig    : W*H u32 path signatures when opts->want_signature, may be NULL.
 * Host pointers. opts->rank/world must be 0/1. */
int  mi_render(mi_ctx* ctx, const mi_camera_desc* cam, const mi_render_opts* opts,
               float* out_rgb_f32, uint8_t* out_rgb_u8, uint32_t* out_sig, mi_stats* stats);

/* Multi-GPU building blocks; all pointers are DEVICE pointers on ctx's device and
 * `stream` is a hipStream_t (NULL = default stream).  Work is queued on `stream`; the
 * megakernel variants never synchronise, the default wavefront variant drives its per-segment
 * iterations from the host and synchronises `stream` inside mi_render_tiles_device.
 *
 * mi_render_tiles_device: rank `opts->rank` of `opts->world` renders its tiles
 *   (tile t -> rank t % world, slot t / world; tiles are numbered row-major over a grid whose row length is the image's tile
 *   columns rounded up to the next integer coprime with `world`, so that every rank meets every column class — the surplus
 *   columns hold no pixel and are written as zeros; mi_compact_size gives the counts) into a compact tile-major buffer
 *   d_compact[tiles_padded][MI_TILE*MI_TILE][3] f32 (row-major inside a tile; pixels
 *   outside the image are written as 0).  d_sig (may be NULL) is [tiles_padded][MI_TILE*MI_TILE] u32.
 * mi_unpermute_device: gathered buffer [world][tiles_padded][1024][3] -> row-major W*H*3 f32.
 * mi_tonemap_device: tracing.rs:244-256 (saturate toward white, gamma, quantise) -> W*H*3 u8. */
int  mi_compact_size(const mi_camera_desc* cam, int32_t world, uint32_t* tiles_total, uint32_t* tiles_padded);
int  mi_render_tiles_device(mi_ctx* ctx, const mi_camera_desc* cam, const mi_render_opts* opts,
                            void* d_compact_f32, void* d_sig_u32, void* stream, mi_stats* stats);
/* Progressive / resumable accumulation (the reference renders all `aa_sample_count` samples of a
 * pixel in one go, tracing.rs:233-241; this splits that loop without changing its result).
 * Traces samples [sample_begin, sample_end) of every pixel of this rank and adds them IN ORDER to the
 * caller-held accumulator d_accum_f32x4[tiles_padded*MI_TILE*MI_TILE] (float4: xyz = running sums,
 * w = bits of the running signature sum).  sample_begin == 0 starts the sums from zero (the buffer
 * need not be cleared).  When sample_end == aa_sample_count the per-pixel means (and signatures) are
 * written to d_compact_f32 / d_sig_u32 exactly as mi_render_tiles_device does; otherwise both may be
 * NULL.  Calls must cover [0, aa_sample_count) in increasing, gap-free order; the accumulator may be
 * copied out and back in between (checkpoint / resume, also into another context with the same
 * scene, camera, seed, rank and world).  The final image is bit-identical to a one-call render.
 * Default (wavefront) path-tracing variant only. */
int  mi_render_samples_device(mi_ctx* ctx, const mi_camera_desc* cam, const mi_render_opts* opts,
                              uint32_t sample_begin, uint32_t sample_end, void* d_accum_f32x4,
                              void* d_compact_f32, void* d_sig_u32, void* stream, mi_stats* stats);
int  mi_unpermute_device(mi_ctx* ctx, const mi_camera_desc* cam, int32_t world,
                         const void* d_gathered_f32, void* d_image_f32, void* stream);
int  mi_tonemap_device(mi_ctx* ctx, const mi_camera_desc* cam,
                       const void* d_image_f32, void* d_image_u8, void* stream);

/* Elapsed time of the most recent path-tracing kernel of this ctx (HIP events on its
 * launch stream); synchronises on the stop event. */
int  mi_last_kernel_ms(mi_ctx* ctx, float* ms);

/* Size and allocate the wavefront pipeline's HBM buffers (path state, sample slots) for
 * this camera with the image shared by `world` ranks, so that the first render does not pay the
 * allocation (about 110 GB for a whole 1080p / 256 spp frame on one GPU).  `max_state_bytes` as in
 * mi_render_opts (0 = the frame in one batch, at most 60 % of the free HBM).  Optional. */
int  mi_reserve(mi_ctx* ctx, const mi_camera_desc* cam, int32_t world, uint64_t max_state_bytes);

/* Wavefront pipeline (MI_VARIANT_WAVEFRONT) of the most recent render: out8 = { sum of wf_main
 * launch durations, of wf_trav, of wf_reduce (ms, HIP events around every launch), launches, sum of wf_trav_f, of
 * wf_replay (the two-stage mesh traversal), sum of the spans of wf_main's class-A parts, 0 }.  A pass after the first launches
 * wf_main in two parts: the class-A blocks run on a second stream BESIDE the walkers of the previous pass (their span includes
 * waiting for CUs the walkers still hold, so it overlaps the wf_trav figure and must not be added to it), the class-B blocks
 * behind the walkers (counted under wf_main). */
int  mi_last_pipeline_ms(mi_ctx* ctx, float* out8);

/* Path counts of the most recent wavefront render, for traffic accounting: out8 = { passes (wf_main launches
 * that left survivors or ended the batch), class-A paths written to (and read back from) the HBM path state summed
 * over the passes, class-B paths likewise, slots the mesh walkers went through (= the class-B paths: the walkers read those lists, there is no queue), sample slots, compact pixels,
 * path segments (every Scene::intersect_ray evaluation, tracing.rs:305, counted on the device by wf_main), samples of dead tiles (tiles from which
 * no camera ray can reach anything: no ray is generated for them — they are black by tracing.rs:306 — and none of the other counts includes them;
 * 0 when signatures were asked for) }.
 * The bytes these stand for (72 B per class-A path and direction, 76 B per class-B path, ...) are in DESIGN.md. */
int  mi_last_pipeline_counts(mi_ctx* ctx, uint64_t* out8);

/* Counters of the most recent *_DIAG launch (synchronises the device):
 * out16 = { A trips, sum of lanes in A trips, interior-step trips, lanes, leaf-step trips,
 *           lanes, B trips, waves, shader-clock cycles in A trips, in B trips, cycles in A's
 *           shade / regenerate / list sections, slab tests inside trees, path segments, 0 }.
 * active-lane fraction of a phase = lanes / (64 * trips). */
int  mi_last_diag(mi_ctx* ctx, uint64_t* out16);

/* Self-test of the kernels' arithmetic shortcuts on this device (about a second).  The kernels compute `1.0 / x` — the
 * reference's Moller-Trumbore `f = 1.0/a` (geometry.rs:340,437), `inv_d` (geometry.rs:57), cgmath's normalize — with a
 * 5-instruction sequence (v_rcp_f32 + one fused Newton step, IEEE division outside the exponent range where that is proven)
 * instead of the 11-instruction IEEE division; this sweeps ALL 2^32 f32 bit patterns and counts the inputs for which the
 * two differ in any bit.  out4 = { mismatches (must be 0), inputs checked, 0, 0 }. */
int  mi_selftest(mi_ctx* ctx, uint64_t* out4);

/* ---- multi-GPU behind the ABI: replaces rayon's row split (tracing.rs:228) with N devices of one node ----
 * One mi_multi owns one context, stream and (per frame) one host thread per device.  The image is cut into MI_TILE^2 tiles,
 * tile t rendered on device t % N; per frame there is exactly ONE exchange — every peer sends its compact tile buffer to
 * device 0 over its own xGMI link (RCCL ncclSend / ncclRecv in one group, resolved with dlopen at mi_multi_create) — then the
 * un-permute and the tone-map run on device 0 — always, so a call with all three output pointers NULL leaves the finished
 * f32 and u8 images resident on device 0 and moves nothing over PCIe.  The image is bit-identical for every N (the RNG is
 * keyed by the global pixel index).  `devices` = NULL means 0 .. n_devices-1.  mi_multi_render: as mi_render; opts->rank / world are ignored;
 * stats->kernel_ms is the slowest device's pipeline pass, stats->total_ms the wall time of the call. */
typedef struct mi_multi mi_multi;
int  mi_multi_create(int n_devices, const int* devices, mi_multi** out);
/* TEST TRANSPORT, not for production use: `n_contexts` ranks that all live on the ONE device `device`, so that everything
 * mi_multi_render does for N >= 2 — one host thread per rank, every rank's pipeline on its own streams, the r * slice receive
 * offsets, the un-permute and tone-map with world > 1, the max-over-ranks statistics — runs on a one-GPU machine.  The only
 * difference from mi_multi_create is the exchange: RCCL is not loaded, and each ncclSend / ncclRecv pair becomes one
 * device-to-device hipMemcpyAsync on the sender's stream that device 0's stream waits for through an event (the ordering
 * the RCCL pair gives).  The ranks share the card, so the call is no faster than mi_render; the image is bit-identical to it.
 * With opts->max_state_bytes == 0 the ranks split the default budget (60 % of the free HBM) evenly. */
int  mi_multi_create_loopback(int n_contexts, int device, mi_multi** out);
void mi_multi_destroy(mi_multi* m);
int  mi_multi_device_count(const mi_multi* m);
/* The context of device number `rank` (0 .. N-1), owned by `m` and valid until mi_multi_destroy: for the per-device queries
 * (mi_last_kernel_ms, mi_last_pipeline_ms, mi_last_pipeline_counts) after a mi_multi_render.  NULL if out of range. */
mi_ctx* mi_multi_context(const mi_multi* m, int rank);
int  mi_multi_scene_upload(mi_multi* m, const mi_scene_desc* scene);
int  mi_multi_reserve(mi_multi* m, const mi_camera_desc* cam, uint64_t max_state_bytes);
int  mi_multi_render(mi_multi* m, const mi_camera_desc* cam, const mi_render_opts* opts,
                     float* out_rgb_f32, uint8_t* out_rgb_u8, uint32_t* out_sig, mi_stats* stats);

/* Thread-local message of the most recent failure in this thread. */
const char* mi_last_error(void);
int  mi_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI_RT_H */
