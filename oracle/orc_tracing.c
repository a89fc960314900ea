/*
 * orc_tracing.c — TEST INFRASTRUCTURE (oracle).  Restates src/util/tracing.rs of the
 * reference: helpers (:54-97), Ray/RayHit (:104-134), Camera::generate_rays (:159-209),
 * Scene::render_to_image (:221-263), background_color (:266-274), phong_shade_ray (:277-297),
 * shade_ray (:300-324, recursive, as written) and the scene hit loop (:326-346).
 * Citations are tracing.rs.
 */
#include <stdlib.h>
#include <pthread.h>
#include "orc_internal.h"

/* ---- helpers :54-97 ---- */
v3 orc_reflect_v(v3 v, v3 n) {                                        /* :54-56  v - 2.0*v.dot(n)*n */
    return v3_sub(v, v3_scale(n, 2.0f * v3_dot(v, n)));
}
float orc_fresnel_v(v3 v, v3 n, float ir) {                           /* :58-62 */
    float r0 = orc_powi2((ir - 1.0f) / (ir + 1.0f));                  /* :60 */
    return r0 + (1.0f - r0) * orc_powi5(1.0f - fabsf(v3_dot(v, n)));  /* :61 */
}
v3 orc_refract_v(v3 v, v3 n, float eta) {                             /* :64-69 */
    float cos_theta = fminf(v3_dot(v3_neg(v), n), 1.0f);              /* :65 */
    v3 r_out_perp = v3_scale(v3_add(v, v3_scale(n, cos_theta)), eta); /* :66 */
    v3 r_out_parallel = v3_scale(n, -sqrtf(fabsf(1.0f - v3_mag2(r_out_perp))));   /* :67 */
    return v3_add(r_out_perp, r_out_parallel);                        /* :68 */
}
v3 orc_rand_sphere_vec(orc_path* p) {                                 /* :71-79 */
    for (;;) {
        v3 dir;
        dir.x = orc_gen_range_m11(&p->rng); dir.y = orc_gen_range_m11(&p->rng); dir.z = orc_gen_range_m11(&p->rng);   /* :74 */
        if (v3_mag2(dir) <= 1.0f) return dir;                         /* :75 */
    }
}
v3 orc_rand_disk_vec(orc_path* p) {                                   /* :81-89 */
    for (;;) {
        v3 dir;
        dir.x = orc_gen_range_m11(&p->rng); dir.y = orc_gen_range_m11(&p->rng); dir.z = 0.0f;   /* :84 */
        if (v3_mag2(dir) <= 1.0f) return dir;                         /* :85 */
    }
}

/* ---- RayHit::new :121-133 ---- */
orc_rayhit orc_rayhit_new(float distance, v3 normal, mi_material material, const orc_ray* ray) {
    orc_rayhit h;
    int frontface = v3_dot(normal, ray->direction) < 0.0f;            /* :122 */
    h.distance = distance;
    h.hitpoint = v3_add(ray->origin, v3_scale(ray->direction, distance));   /* :125 */
    h.normal = frontface ? normal : v3_neg(normal);                   /* :126 */
    h.material = material;
    h.frontface = frontface;
    h.has_tex_coords = 0; h.tex_coords = v2_make(0.0f, 0.0f);
    h.has_tangent = 0; h.tangent = v3_zero();
    h.has_bitangent = 0; h.bitangent = v3_zero();
    h.object = -1;
    return h;
}

/* ---- Camera::generate_rays :159-209, one sample i of pixel (screen_x, screen_y).
 * The reference builds the whole Vec<Ray> for the pixel from one thread_rng and then
 * shades it; here each sample owns a stream, whose first draws are the ones below. */
static orc_ray generate_ray(const mi_camera_desc* cam, uint32_t screen_x, uint32_t screen_y, uint32_t i, orc_path* p) {
    float pixel_size = 1.0f / (float)cam->screen_height;              /* :160 */
    float n = (float)cam->aa_sample_count;                            /* :162 */
    float rootn = sqrtf(n);                                           /* :163 */
    float rand_x = (float)orc_gen_range_u32(&p->rng, cam->aa_sample_count);   /* :167 */
    float rand_y = (float)orc_gen_range_u32(&p->rng, cam->aa_sample_count);   /* :168 */
    float subpixel_x = (float)(i / (uint32_t)rootn);                  /* :169 */
    float subpixel_y = (float)(i % (uint32_t)rootn);                  /* :170 */
    float off_x = (subpixel_x - 0.5f * rootn) * pixel_size / rootn + (rand_x - 0.5f * n) * pixel_size / n;   /* :172 */
    float off_y = (subpixel_y - 0.5f * rootn) * pixel_size / rootn + (rand_y - 0.5f * n) * pixel_size / n;   /* :173 */
    v3 cam_space_pixel_center = v3_make(                              /* :177-181 */
        pixel_size * ((float)screen_x - 0.5f * (float)cam->screen_width + 0.5f) + off_x,
        pixel_size * (0.5f + 0.5f * (float)cam->screen_height - (float)screen_y) + off_y,
        -cam->focal_length);
    v3 focus_plane_pixel_center = v3_scale(v3_normalize(cam_space_pixel_center), cam->focus_dist);   /* :183 */
    v3 lens_origin = v3_scale(orc_rand_disk_vec(p), cam->lens_radius);                               /* :184 */
    v3 view_dir = v3_from(cam->view_dir), up = v3_from(cam->up);
    m3 rotation;                                                      /* :187-191 */
    rotation.c0 = v3_normalize(v3_cross(view_dir, up));
    rotation.c1 = up;
    rotation.c2 = v3_neg(view_dir);
    orc_ray ray;
    if (cam->projection_mode == MI_PROJ_ORTHOGRAPHIC) {
        /* :196,200 — the origin stays in camera space (eyepoint and rotation are NOT applied) and the
         * direction is view_dir, which :204 then rotates like any camera-space direction */
        ray.origin = v3_make(cam_space_pixel_center.x, cam_space_pixel_center.y, 0.0f);
        ray.direction = view_dir;
    } else {
        ray.origin = v3_add(v3_from(cam->eyepoint), m3_mul_v3(rotation, lens_origin));   /* :197 */
        ray.direction = v3_normalize(v3_sub(focus_plane_pixel_center, lens_origin));     /* :201 */
    }
    ray.direction = m3_mul_v3(rotation, ray.direction);               /* :204 */
    return ray;
}

/* ---- impl Intersectable for Scene :327-346 ---- */
int orc_scene_intersect_ray(const orc_scene* s, const orc_ray* ray, float t_min, float t_max, orc_path* p, orc_rayhit* out) {
    int have_best = 0;                                                /* :329 */
    orc_rayhit best_hit;
    if (p->cnt) p->cnt->segments++;
    for (int i = 0; i < s->n_objects; i++) {                          /* :330 */
        orc_rayhit hit; int got = 0;
        const mi_object* o = &s->objects[i];
        if (p->cnt) p->cnt->object_tests++;
        switch (o->kind) {                                            /* dyn dispatch :331 */
        case MI_OBJ_SPHERE:   got = orc_sphere_intersect(s, &s->spheres[o->index], ray, t_min, t_max, &hit); break;
        case MI_OBJ_TRIANGLE: got = orc_triangle_intersect(s, &s->triangles[o->index], ray, t_min, t_max, &hit); break;
        case MI_OBJ_PLANE:    got = orc_plane_intersect(s, &s->planes[o->index], ray, t_min, t_max, &hit); break;
        case MI_OBJ_VOLUME:   got = orc_volume_intersect(s, &s->volumes[o->index], ray, t_min, t_max, p, &hit); break;
        case MI_OBJ_MESH:     got = orc_mesh_intersect(s, &s->meshes[o->index], ray, t_min, t_max, p, &hit); break;
        default: break;
        }
        if (got) {
            hit.object = i;
            if (!have_best) { best_hit = hit; have_best = 1; }        /* :333 */
            else if (hit.distance < best_hit.distance) best_hit = hit;   /* :335-336 */
        }
    }
    if (have_best) *out = best_hit;
    return have_best;                                                 /* :345 */
}

/* ---- background_color :266-274 ---- */
static v3 background_color(v3 dir) { (void)dir; return v3_zero(); }

/* path-signature steps (DESIGN.md "Path signature"; diagnostic, not part of the reference) */
static inline void sig_hit(orc_path* p, const orc_rayhit* h) {
    uint32_t tb; memcpy(&tb, &h->distance, 4);
    p->sig = orc_lowbias32((p->sig ^ tb) + (uint32_t)(h->object + 1) * 0x9e3779b1u);
}
/* a path that leaves the scene folds the final RNG state in (it pins every draw made so
 * far); a path cut by path_depth does not, because the draws of its last scatter
 * (tracing.rs:312 at depth-1) can no longer influence the image and an implementation
 * may skip that scatter */
static inline void sig_end_miss(orc_path* p) {
    p->sig = orc_lowbias32(p->sig ^ p->rng.s0 ^ orc_rotl32(p->rng.s1, 16));
}
static inline void sig_end_depth(orc_path* p) {
    p->sig = orc_lowbias32(p->sig ^ 0x5bd1e995u);
}

/* ---- Scene::shade_ray :300-324 (recursive, as the reference) ---- */
static v3 shade_ray(const orc_scene* s, const mi_camera_desc* cam, const orc_ray* ray, uint32_t recursion_depth, orc_path* p) {
    if (recursion_depth >= cam->path_depth) {                         /* :301 */
        sig_end_depth(p);
        return background_color(ray->direction);                      /* :302 */
    }
    orc_rayhit hit;
    if (!orc_scene_intersect_ray(s, ray, 0.001f, cam->max_trace_dist, p, &hit)) {   /* :305 */
        sig_end_miss(p);
        return background_color(ray->direction);                      /* :306 */
    }
    sig_hit(p, &hit);
    v3 integral = v3_zero();                                          /* :309 */
    for (uint32_t i = 0; i < cam->path_samples; i++) {                /* :310 */
        orc_ray new_ray; v3 brdf_term; float pdf;
        orc_material_scatter(&hit.material, &hit, ray, p, &new_ray, &brdf_term, &pdf);   /* :312 */
        float dot_term = (v3_mag2(hit.normal) > 0.0f)                 /* :313 */
            ? orc_clampf(fabsf(v3_dot(new_ray.direction, hit.normal)), 0.0f, 1.0f) : 1.0f;
        v3 incoming_light = shade_ray(s, cam, &new_ray, recursion_depth + 1, p);         /* :314 */
        /* integral += (dot_term*(brdf_term.mul_element_wise(incoming_light))) / pdf      :316 */
        integral = v3_add(integral, v3_divs(v3_scale(v3_mul_elem(brdf_term, incoming_light), dot_term), pdf));
    }
    integral = v3_divs(integral, (float)cam->path_samples);           /* :318 */
    return v3_add(orc_material_emission(&hit.material), integral);    /* :321 */
}

/* ---- Scene::phong_shade_ray :277-297 (ShadingMode::Phong, "usually just used for debugging") ---- */
static v3 phong_shade_ray(const orc_scene* s, const mi_camera_desc* cam, const orc_ray* ray, orc_path* p) {
    orc_rayhit hit;
    if (!orc_scene_intersect_ray(s, ray, 0.0f, cam->max_trace_dist, p, &hit)) {          /* :279 */
        sig_end_miss(p);
        return background_color(ray->direction);                      /* :280 */
    }
    sig_hit(p, &hit);
    v3 light = v3_from(s->point_light_pos);
    v3 to_light = v3_normalize(v3_sub(light, hit.hitpoint));          /* :283 */
    v3 to_camera = v3_normalize(v3_sub(v3_from(cam->eyepoint), hit.hitpoint));           /* :284 */
    /* :285  -to_light + 2.0*dot(to_light, hit.normal)*hit.normal */
    v3 reflected = v3_add(v3_neg(to_light), v3_scale(hit.normal, 2.0f * v3_dot(to_light, hit.normal)));
    float diffuse_weight = orc_clampf(v3_dot(hit.normal, to_light), 0.0f, 1.0f);         /* :286 */
    float specular_weight = orc_pow40(orc_clampf(v3_dot(to_camera, reflected), 0.0f, 1.0f));   /* :287 */
    orc_ray shadow_ray;                                               /* :289 */
    shadow_ray.origin = v3_add(hit.hitpoint, v3_scale(hit.normal, 0.01f));
    shadow_ray.direction = to_light;
    float shadow_weight = 1.0f;                                       /* :290-293 */
    orc_rayhit sh;
    if (orc_scene_intersect_ray(s, &shadow_ray, 0.0f, v3_mag(v3_sub(light, hit.hitpoint)), p, &sh)) {
        sig_hit(p, &sh);
        /* the arm's `hit` shadows the outer one: both distances are the SHADOW hit's */
        shadow_weight = (sh.distance * sh.distance > v3_mag2(v3_sub(light, sh.hitpoint))) ? 1.0f : 0.3f;
    }
    orc_ray new_ray; v3 brdf_term; float pdf;                         /* :294 `hit.material.scatter(&hit, ray).1` */
    orc_material_scatter(&hit.material, &hit, ray, p, &new_ray, &brdf_term, &pdf);
    v3 amb = v3_from(s->ambient);
    v3 sum = v3_add(v3_add(amb, v3_scale(brdf_term, diffuse_weight)), v3_scale(v3_make(0.4f, 0.4f, 0.4f), specular_weight));
    return v3_scale(sum, shadow_weight);
}

/* ---- pixel epilogue :244-256 ---- */
static void tonemap_pixel(v3 final_color, float gamma, uint8_t* out) {
    float tmp[3] = { final_color.x, final_color.y, final_color.z };   /* :244 */
    float fc[3]  = { final_color.x, final_color.y, final_color.z };
    for (int i = 0; i < 3; i++) {                                     /* :245-251 */
        float d = tmp[i] - 1.0f;
        if (d > 0.0f) { fc[(i + 1) % 3] += d; fc[(i + 2) % 3] += d; }
    }
    for (int i = 0; i < 3; i++) {                                     /* :254-256; `as u8` saturates */
        float q = powf(orc_clampf(fc[i], 0.0f, 1.0f), 1.0f / gamma) * 255.9999f;
        out[i] = (q != q) ? 0 : (q <= 0.0f ? 0 : (q >= 255.0f ? 255 : (uint8_t)q));
    }
}

/* one pixel of render_to_image :229-257 */
static void render_pixel(const orc_scene* s, const mi_camera_desc* cam, uint32_t seed, uint32_t x, uint32_t y,
                         float* out_f32, uint8_t* out_u8, uint32_t* out_sig, orc_counters* cnt) {
    v3 final_color = v3_zero();                                       /* :232 */
    uint32_t n = cam->aa_sample_count;
    uint32_t pixel = y * cam->screen_width + x;
    uint32_t sigsum = 0;
    for (uint32_t i = 0; i < n; i++) {                                /* :233 */
        orc_path p; p.cnt = cnt; p.sig = 0;
        orc_rng_init(&p.rng, seed, pixel, i);
        orc_ray ray = generate_ray(cam, x, y, i, &p);                 /* :231 */
        if (cam->shading_mode == MI_SHADE_PHONG)                      /* :234-239 */
            final_color = v3_add(final_color, phong_shade_ray(s, cam, &ray, &p));
        else
            final_color = v3_add(final_color, shade_ray(s, cam, &ray, 0, &p));
        sigsum += p.sig;
        if (cnt) { cnt->samples++; cnt->rng_draws += p.rng.draws; }
    }
    final_color = v3_divs(final_color, (float)n);                     /* :241 */
    if (out_f32) { out_f32[0] = final_color.x; out_f32[1] = final_color.y; out_f32[2] = final_color.z; }
    if (out_u8) tonemap_pixel(final_color, cam->gamma, out_u8);
    if (out_sig) *out_sig = sigsum;
}

/* ---- Scene::render_to_image :221-263: rows are the parallel unit (:228) ---- */
typedef struct {
    const orc_scene* s; const mi_camera_desc* cam; uint32_t seed;
    int x0, y0, w, h, row_stride;
    float* out_f32; uint8_t* out_u8; uint32_t* out_sig;
    int* next_row; pthread_mutex_t* lock;
    orc_counters cnt; int want_cnt;
} render_job;

static void* render_worker(void* arg) {
    render_job* j = (render_job*)arg;
    for (;;) {
        pthread_mutex_lock(j->lock);
        int row = (*j->next_row)++;
        pthread_mutex_unlock(j->lock);
        if (row >= j->h) break;
        for (int cx = 0; cx < j->w; cx++) {                           /* :229 */
            size_t k = (size_t)row * j->w + cx;
            render_pixel(j->s, j->cam, j->seed, (uint32_t)(j->x0 + cx), (uint32_t)(j->y0 + row * j->row_stride),
                         j->out_f32 ? j->out_f32 + 3 * k : NULL, j->out_u8 ? j->out_u8 + 3 * k : NULL,
                         j->out_sig ? j->out_sig + k : NULL, j->want_cnt ? &j->cnt : NULL);
        }
    }
    return NULL;
}

static int check_camera(const mi_camera_desc* cam) {
    if (!cam) return MI_ERR_INVALID;
    if (cam->projection_mode != MI_PROJ_PERSPECTIVE && cam->projection_mode != MI_PROJ_ORTHOGRAPHIC) return MI_ERR_INVALID;
    if (cam->shading_mode != MI_SHADE_PATHTRACE && cam->shading_mode != MI_SHADE_PHONG) return MI_ERR_INVALID;
    if (cam->screen_width == 0 || cam->screen_height == 0 || cam->aa_sample_count == 0 || cam->path_samples == 0) return MI_ERR_INVALID;
    uint32_t r = (uint32_t)sqrtf((float)cam->aa_sample_count);
    if (r == 0) return MI_ERR_INVALID;
    return MI_OK;
}

int orc_render(const orc_scene* s, const mi_camera_desc* cam, uint32_t seed, int n_threads,
               int x0, int y0, int w, int h, int row_stride,
               float* out_rgb_f32, uint8_t* out_rgb_u8, uint32_t* out_sig, orc_counters* counters) {
    int rc = check_camera(cam);
    if (rc != MI_OK || !s) return rc != MI_OK ? rc : MI_ERR_INVALID;
    if (row_stride < 1) return MI_ERR_INVALID;
    if (x0 < 0 || y0 < 0 || w < 0 || h < 0 || (uint32_t)(x0 + w) > cam->screen_width ||
        (h > 0 && (uint32_t)(y0 + (h - 1) * row_stride) >= cam->screen_height))
        return MI_ERR_INVALID;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    int next_row = 0;
    pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
    render_job* jobs = (render_job*)calloc((size_t)n_threads, sizeof(render_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        render_job* j = &jobs[t];
        j->s = s; j->cam = cam; j->seed = seed; j->x0 = x0; j->y0 = y0; j->w = w; j->h = h; j->row_stride = row_stride;
        j->out_f32 = out_rgb_f32; j->out_u8 = out_rgb_u8; j->out_sig = out_sig;
        j->next_row = &next_row; j->lock = &lock; j->want_cnt = counters != NULL;
    }
    if (n_threads == 1) render_worker(&jobs[0]);
    else {
        for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, render_worker, &jobs[t]);
        for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
    }
    if (counters) {
        orc_counters c; memset(&c, 0, sizeof c);
        for (int t = 0; t < n_threads; t++) {
            const orc_counters* q = &jobs[t].cnt;
            c.samples += q->samples; c.segments += q->segments; c.object_tests += q->object_tests;
            c.mesh_tests += q->mesh_tests; c.mesh_entered += q->mesh_entered; c.mesh_hits += q->mesh_hits;
            c.box_tests += q->box_tests; c.tri_tests += q->tri_tests; c.texel_fetches += q->texel_fetches;
            c.rng_draws += q->rng_draws;
        }
        *counters = c;
    }
    free(jobs); free(th);
    return MI_OK;
}

/* ---- scene construction ---- */
static void* dup_mem(const void* p, size_t n) {
    if (!p || n == 0) return NULL;
    void* q = malloc(n); memcpy(q, p, n); return q;
}

int orc_scene_create(const mi_scene_desc* d, orc_scene** out) {
    if (!d || !out) return MI_ERR_INVALID;
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    memcpy(s->point_light_pos, d->point_light_pos, sizeof s->point_light_pos);
    memcpy(s->ambient, d->ambient, sizeof s->ambient);
    s->n_objects = d->n_objects;     s->objects = (mi_object*)dup_mem(d->objects, sizeof(mi_object) * (size_t)d->n_objects);
    s->n_spheres = d->n_spheres;     s->spheres = (mi_sphere*)dup_mem(d->spheres, sizeof(mi_sphere) * (size_t)d->n_spheres);
    s->n_triangles = d->n_triangles; s->triangles = (mi_triangle*)dup_mem(d->triangles, sizeof(mi_triangle) * (size_t)d->n_triangles);
    s->n_planes = d->n_planes;       s->planes = (mi_plane*)dup_mem(d->planes, sizeof(mi_plane) * (size_t)d->n_planes);
    s->n_volumes = d->n_volumes;     s->volumes = (mi_volume*)dup_mem(d->volumes, sizeof(mi_volume) * (size_t)d->n_volumes);
    s->n_materials = d->n_materials; s->materials = (mi_material*)dup_mem(d->materials, sizeof(mi_material) * (size_t)d->n_materials);
    s->n_boundary_objects = d->boundary_objects ? d->n_boundary_objects : 0;
    s->boundary_objects = (mi_object*)dup_mem(d->boundary_objects, sizeof(mi_object) * (size_t)s->n_boundary_objects);
    s->n_textures = d->n_textures;   s->textures = (mi_texture*)dup_mem(d->textures, sizeof(mi_texture) * (size_t)d->n_textures);
    for (int i = 0; i < s->n_textures; i++)
        s->textures[i].rgb = (const uint8_t*)dup_mem(d->textures[i].rgb, (size_t)d->textures[i].width * d->textures[i].height * 3);
    s->n_meshes = d->n_meshes;
    s->meshes = (orc_mesh*)calloc((size_t)(d->n_meshes > 0 ? d->n_meshes : 1), sizeof(orc_mesh));
    for (int i = 0; i < d->n_meshes; i++) {
        const mi_mesh* m = &d->meshes[i]; orc_mesh* o = &s->meshes[i];
        if (!m->positions || !m->normals || !m->texcoords || !m->indices || m->n_triangles < 1) { orc_scene_destroy(s); return MI_ERR_INVALID; }
        o->n_vertices = m->n_vertices; o->n_triangles = m->n_triangles;
        o->positions = (float*)dup_mem(m->positions, sizeof(float) * 3 * (size_t)m->n_vertices);
        o->normals   = (float*)dup_mem(m->normals,   sizeof(float) * 3 * (size_t)m->n_vertices);
        o->texcoords = (float*)dup_mem(m->texcoords, sizeof(float) * 2 * (size_t)m->n_vertices);
        o->indices   = (uint32_t*)dup_mem(m->indices, sizeof(uint32_t) * 3 * (size_t)m->n_triangles);
        memcpy(o->transform, m->transform, sizeof o->transform);
        memcpy(o->inv_transform, m->inv_transform, sizeof o->inv_transform);
        o->material = m->material;
        memcpy(o->textures, m->textures, sizeof o->textures);
        for (int k = 0; k < 3 * m->n_triangles; k++)
            if (o->indices[k] >= (uint32_t)m->n_vertices) { orc_scene_destroy(s); return MI_ERR_INVALID; }
        orc_mesh_build_bvh(o);                                        /* geometry.rs:170 */
    }
    /* index validation (the reference cannot express a dangling Arc) */
    for (int i = 0; i < s->n_objects; i++) {
        const mi_object* o = &s->objects[i]; int n = 0;
        switch (o->kind) {
        case MI_OBJ_SPHERE: n = s->n_spheres; break;   case MI_OBJ_TRIANGLE: n = s->n_triangles; break;
        case MI_OBJ_PLANE: n = s->n_planes; break;     case MI_OBJ_VOLUME: n = s->n_volumes; break;
        case MI_OBJ_MESH: n = s->n_meshes; break;      default: n = 0; break;
        }
        if (o->index < 0 || o->index >= n) { orc_scene_destroy(s); return MI_ERR_INVALID; }
    }
    for (int i = 0; i < s->n_volumes; i++) {                           /* ConvexVolume.boundary: Arc<dyn Intersectable> */
        const mi_volume* v = &s->volumes[i];
        int first = 0, count = 1; const mi_object* entries = NULL; mi_object one;
        if (v->boundary_kind == MI_OBJ_SPHERE) continue;                /* the inline sphere */
        if (v->boundary_kind == MI_OBJ_SCENE) {
            first = v->boundary_index; count = v->boundary_count; entries = s->boundary_objects;
            if (first < 0 || count < 0 || (int64_t)first + (int64_t)count > (int64_t)s->n_boundary_objects) { orc_scene_destroy(s); return MI_ERR_INVALID; }
        } else { one.kind = v->boundary_kind; one.index = v->boundary_index; entries = &one; }
        for (int k = 0; k < count; k++) {
            const mi_object* e = &entries[first + k]; int n = -1;
            switch (e->kind) {
            case MI_OBJ_SPHERE: n = s->n_spheres; break;   case MI_OBJ_TRIANGLE: n = s->n_triangles; break;
            case MI_OBJ_PLANE: n = s->n_planes; break;     case MI_OBJ_MESH: n = s->n_meshes; break;
            default: orc_scene_destroy(s); return MI_ERR_UNSUPPORTED;   /* a volume or a scene inside a boundary */
            }
            if (e->index < 0 || e->index >= n) { orc_scene_destroy(s); return MI_ERR_INVALID; }
        }
    }
    *out = s;
    return MI_OK;
}

void orc_scene_destroy(orc_scene* s) {
    if (!s) return;
    for (int i = 0; i < s->n_meshes; i++) {
        orc_mesh* o = &s->meshes[i];
        free(o->positions); free(o->normals); free(o->texcoords); free(o->indices);
        orc_bvh_free(o->bvh_root);
    }
    for (int i = 0; i < s->n_textures; i++) free((void*)s->textures[i].rgb);
    free(s->meshes); free(s->objects); free(s->spheres); free(s->triangles); free(s->planes);
    free(s->volumes); free(s->materials); free(s->textures); free(s->boundary_objects);
    free(s);
}

/* ---- unit-level exports ---- */
int orc_intersect(const orc_scene* s, const float origin[3], const float dir[3], float t_min, float t_max,
                  uint32_t seed, uint32_t pixel, uint32_t sample, orc_hit_rec* out) {
    if (!s || !out) return MI_ERR_INVALID;
    orc_path p; p.cnt = NULL; p.sig = 0; orc_rng_init(&p.rng, seed, pixel, sample);
    orc_ray ray; ray.origin = v3_from(origin); ray.direction = v3_from(dir);
    orc_rayhit h;
    memset(out, 0, sizeof *out);
    if (!orc_scene_intersect_ray(s, &ray, t_min, t_max, &p, &h)) { out->hit = 0; out->object = -1; return MI_OK; }
    out->hit = 1; out->distance = h.distance;
    out->hitpoint[0] = h.hitpoint.x; out->hitpoint[1] = h.hitpoint.y; out->hitpoint[2] = h.hitpoint.z;
    out->normal[0] = h.normal.x; out->normal[1] = h.normal.y; out->normal[2] = h.normal.z;
    out->frontface = h.frontface; out->object = h.object; out->material = h.material;
    out->has_uv = h.has_tex_coords; out->uv[0] = h.tex_coords.x; out->uv[1] = h.tex_coords.y;
    return MI_OK;
}

int orc_generate_rays(const mi_camera_desc* cam, uint32_t seed, uint32_t x, uint32_t y, float* out) {
    int rc = check_camera(cam); if (rc != MI_OK) return rc;
    for (uint32_t i = 0; i < cam->aa_sample_count; i++) {
        orc_path p; p.cnt = NULL; p.sig = 0; orc_rng_init(&p.rng, seed, y * cam->screen_width + x, i);
        orc_ray r = generate_ray(cam, x, y, i, &p);
        out[6 * i + 0] = r.origin.x; out[6 * i + 1] = r.origin.y; out[6 * i + 2] = r.origin.z;
        out[6 * i + 3] = r.direction.x; out[6 * i + 4] = r.direction.y; out[6 * i + 5] = r.direction.z;
    }
    return MI_OK;
}

int orc_shade(const orc_scene* s, const mi_camera_desc* cam, const float origin[3], const float dir[3],
              uint32_t seed, uint32_t pixel, uint32_t sample, float out_rgb[3]) {
    int rc = check_camera(cam); if (rc != MI_OK) return rc;
    orc_path p; p.cnt = NULL; p.sig = 0; orc_rng_init(&p.rng, seed, pixel, sample);
    orc_ray ray; ray.origin = v3_from(origin); ray.direction = v3_from(dir);
    v3 c = (cam->shading_mode == MI_SHADE_PHONG) ? phong_shade_ray(s, cam, &ray, &p) : shade_ray(s, cam, &ray, 0, &p);
    out_rgb[0] = c.x; out_rgb[1] = c.y; out_rgb[2] = c.z;
    return MI_OK;
}

int orc_scatter(const mi_material* m, const float hitpoint[3], const float normal[3], int frontface,
                const float ray_dir[3], uint32_t seed, uint32_t pixel, uint32_t sample, float out[7]) {
    orc_path p; p.cnt = NULL; p.sig = 0; orc_rng_init(&p.rng, seed, pixel, sample);
    orc_rayhit h; memset(&h, 0, sizeof h);
    h.hitpoint = v3_from(hitpoint); h.normal = v3_from(normal); h.frontface = frontface; h.material = *m;
    orc_ray ray; ray.origin = v3_zero(); ray.direction = v3_from(ray_dir);
    orc_ray nr; v3 brdf; float pdf;
    orc_material_scatter(m, &h, &ray, &p, &nr, &brdf, &pdf);
    out[0] = nr.direction.x; out[1] = nr.direction.y; out[2] = nr.direction.z;
    out[3] = brdf.x; out[4] = brdf.y; out[5] = brdf.z; out[6] = pdf;
    return MI_OK;
}

void orc_reflect(const float v[3], const float n[3], float out[3]) {
    v3 r = orc_reflect_v(v3_from(v), v3_from(n)); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_refract(const float v[3], const float n[3], float eta, float out[3]) {
    v3 r = orc_refract_v(v3_from(v), v3_from(n), eta); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_fresnel(const float v[3], const float n[3], float ir) { return orc_fresnel_v(v3_from(v), v3_from(n), ir); }
void orc_between_vectors_mat(const float a[3], const float b[3], float out9[9]) {
    m3 m = orc_between_vectors(v3_from(a), v3_from(b));
    out9[0] = m.c0.x; out9[1] = m.c0.y; out9[2] = m.c0.z;
    out9[3] = m.c1.x; out9[4] = m.c1.y; out9[5] = m.c1.z;
    out9[6] = m.c2.x; out9[7] = m.c2.y; out9[8] = m.c2.z;
}
float orc_logf_export(float x) { return orc_logf(x); }
void orc_tonemap_pixel(const float rgb[3], float gamma, uint8_t out[3]) { tonemap_pixel(v3_from(rgb), gamma, out); }
void orc_rng_words(uint32_t seed, uint32_t pixel, uint32_t sample, int n, uint32_t* out) {
    orc_rng r; orc_rng_init(&r, seed, pixel, sample);
    for (int i = 0; i < n; i++) out[i] = orc_next_u32(&r);
}

static void bvh_stats_rec(const orc_bvhnode* n, int d, int* nodes, int* depth, int* flat) {
    (*nodes)++;
    if (d > *depth) *depth = d;
    if (!n->has_primitive) {
        if (n->aabb.min.x == n->aabb.max.x || n->aabb.min.y == n->aabb.max.y || n->aabb.min.z == n->aabb.max.z) (*flat)++;
        if (n->left) bvh_stats_rec(n->left, d + 1, nodes, depth, flat);
        if (n->right) bvh_stats_rec(n->right, d + 1, nodes, depth, flat);
    }
}
int orc_bvh_stats(const orc_scene* s, int mesh, int* nodes, int* depth, int* flat_inner) {
    if (!s || mesh < 0 || mesh >= s->n_meshes) return MI_ERR_INVALID;
    *nodes = 0; *depth = 0; *flat_inner = 0;
    bvh_stats_rec(s->meshes[mesh].bvh_root, 0, nodes, depth, flat_inner);
    return MI_OK;
}
