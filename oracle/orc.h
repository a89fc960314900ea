/*
 * orc.h — public interface of the ORACLE (liborc.so).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE.  A plain-C, CPU, literal restatement of the
 * hot path of mbk6/CS397RayTracingSP22 (recursive shade_ray, recursive pointer BVH,
 * a full RayHit per candidate, exactly as the Rust source is structured).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / the timed CPU baseline.  The product (libmi_rt.so) never links
 * or calls it and fails loudly when its HIP code is missing.
 *
 * PARITY UNPINNED: the reference holds no test, golden vector, known-answer value or
 * fixture for this path (no #[test] anywhere), it cannot be built here
 * (no cargo/rustc, crates not vendored) and its RNG is unseeded (thread_rng), so no
 * output of the real program can be reproduced bit for bit.  What pins this oracle:
 * line-by-line restatement with file:line citations, analytic known-answer tests
 * (tests/test_oracle_*.py) and the closed-form "furnace" identity of SURVEY.md §4.
 *
 * It consumes the same POD scene description as the product (include/mi_rt.h is an
 * interface header, not product code).
 */
#ifndef ORC_H
#define ORC_H

#include "../include/mi_rt.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_scene orc_scene;

/* work counters, summed over all samples of a render (fixtures for DESIGN.md's
 * algorithmic-bytes figure) */
typedef struct orc_counters {
    uint64_t samples;        /* camera rays                                     */
    uint64_t segments;       /* Scene::intersect_ray evaluations  tracing.rs:305 */
    uint64_t object_tests;   /* object.intersect_ray calls        tracing.rs:331 */
    uint64_t mesh_tests;     /* StaticMesh::intersect_ray calls   geometry.rs:301 */
    uint64_t mesh_entered;   /* ... whose root AABB test passed                   */
    uint64_t mesh_hits;      /* ... that returned Some(hit)                       */
    uint64_t box_tests;      /* AABB::intersect_ray calls         geometry.rs:52  */
    uint64_t tri_tests;      /* IndexedTriangle::intersect_ray    geometry.rs:331 */
    uint64_t texel_fetches;  /* Texture::sample calls             texture.rs:26   */
    uint64_t rng_draws;      /* next_u32 calls                                    */
} orc_counters;

/* Build the oracle's scene: pointer BVH per mesh with the reference's topology
 * (geometry.rs:175-217).  The description is deep-copied. */
int  orc_scene_create(const mi_scene_desc* desc, orc_scene** out);
void orc_scene_destroy(orc_scene* s);

/* Scene::render_to_image (tracing.rs:221-263) over the window [x0,x0+w) x [y0,y0+h)
 * of the cam->screen_width x cam->screen_height image (the RNG is keyed by the global
 * pixel index, so a window equals the same region of the whole image).
 *   out_rgb_f32 : w*h*3 per-pixel mean radiance BEFORE saturation/gamma (may be NULL)
 *   out_rgb_u8  : w*h*3 bytes as tracing.rs:254-256 writes them      (may be NULL)
 *   out_sig     : w*h path signatures (DESIGN.md "Path signature")   (may be NULL)
 *   n_threads   : row-parallel workers (mirrors rayon's per-row tasks, tracing.rs:228)
 *   row_stride  : window row k is image row y0 + k*row_stride (1 = contiguous window);
 *                 > 1 samples the whole image height with few rows (bench.py cpu_baseline) */
int  orc_render(const orc_scene* s, const mi_camera_desc* cam, uint32_t seed, int n_threads,
                int x0, int y0, int w, int h, int row_stride,
                float* out_rgb_f32, uint8_t* out_rgb_u8, uint32_t* out_sig, orc_counters* counters);

/* ---- unit-level entry points for known-answer tests ---- */
typedef struct orc_hit_rec {
    int32_t hit;             /* 0 = None */
    float   distance;
    float   hitpoint[3];
    float   normal[3];
    int32_t frontface;
    int32_t object;          /* index into Scene.objects */
    mi_material material;    /* resolved material at the hit */
    float   uv[2];
    int32_t has_uv;
} orc_hit_rec;

/* Scene::intersect_ray (tracing.rs:326-346) for one ray with its own RNG stream. */
int  orc_intersect(const orc_scene* s, const float origin[3], const float dir[3],
                   float t_min, float t_max, uint32_t seed, uint32_t pixel, uint32_t sample,
                   orc_hit_rec* out);
/* Camera::generate_rays (tracing.rs:159-209): rays for all aa samples of one pixel, out[n][6]. */
int  orc_generate_rays(const mi_camera_desc* cam, uint32_t seed, uint32_t x, uint32_t y, float* out);
/* Scene::shade_ray (tracing.rs:300-324) for one ray. */
int  orc_shade(const orc_scene* s, const mi_camera_desc* cam, const float origin[3], const float dir[3],
               uint32_t seed, uint32_t pixel, uint32_t sample, float out_rgb[3]);
/* Material::scatter (materials.rs) on a synthetic hit: out = dir[3], brdf[3], pdf. */
int  orc_scatter(const mi_material* m, const float hitpoint[3], const float normal[3], int frontface,
                 const float ray_dir[3], uint32_t seed, uint32_t pixel, uint32_t sample, float out[7]);
/* helpers (tracing.rs:54-69), Texture::sample, between_vectors, logf, tone-map of one pixel */
void orc_reflect(const float v[3], const float n[3], float out[3]);
void orc_refract(const float v[3], const float n[3], float eta, float out[3]);
float orc_fresnel(const float v[3], const float n[3], float ir);
void orc_texture_sample(const mi_texture* t, float u, float v, float out[3]);
void orc_between_vectors_mat(const float a[3], const float b[3], float out9[9]);
float orc_logf_export(float x);
void orc_tonemap_pixel(const float rgb[3], float gamma, uint8_t out[3]);
/* raw RNG stream, n words */
void orc_rng_words(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out);
/* BVH topology probe: node count, depth, flat (zero-extent) interior nodes of mesh m */
int  orc_bvh_stats(const orc_scene* s, int mesh, int* nodes, int* depth, int* flat_inner);

#ifdef __cplusplus
}
#endif
#endif
