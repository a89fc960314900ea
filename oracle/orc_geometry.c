/*
 * orc_geometry.c — TEST INFRASTRUCTURE (oracle).  Restates src/util/geometry.rs of the
 * reference: AABB slab test, recursive BVH, StaticMesh, IndexedTriangle, Sphere,
 * Triangle, Plane, ConvexVolume.  Structure and evaluation order follow the Rust
 * source line by line (citations are geometry.rs unless noted); nothing is hoisted,
 * cached or reordered.
 */
#include <stdlib.h>
#include "orc_internal.h"

/* ---- AABB (geometry.rs:22-83) ---- */

/* AABB::aabb_surrounding  :28-41 */
static orc_aabb aabb_surrounding(const orc_aabb* a, const orc_aabb* b) {
    orc_aabb r;
    r.min = v3_make(fminf(a->min.x, b->min.x), fminf(a->min.y, b->min.y), fminf(a->min.z, b->min.z));
    r.max = v3_make(fmaxf(a->max.x, b->max.x), fmaxf(a->max.y, b->max.y), fmaxf(a->max.z, b->max.z));
    return r;
}

/* impl Intersectable for AABB :52-79.  Returns 1 for Some(dummy hit), 0 for None.
 * f32::max / f32::min return the non-NaN operand (= fmaxf / fminf). */
int orc_aabb_intersect(const orc_aabb* b, const orc_ray* ray, float t_min, float t_max, orc_path* p) {
    if (p && p->cnt) p->cnt->box_tests++;
    float tmin = t_min;                                               /* :54 */
    float tmax = t_max;                                               /* :55 */
    for (int axis = 0; axis < 3; axis++) {                            /* :56 */
        float inv_d = 1.0f / v3_get(ray->direction, axis);            /* :57 */
        float t0 = (v3_get(b->min, axis) - v3_get(ray->origin, axis)) * inv_d;   /* :58 */
        float t1 = (v3_get(b->max, axis) - v3_get(ray->origin, axis)) * inv_d;   /* :59 */
        if (inv_d < 0.0f) { float t = t0; t0 = t1; t1 = t; }          /* :60-62 */
        tmin = fmaxf(t0, tmin);                                       /* :63 */
        tmax = fminf(t1, tmax);                                       /* :64 */
        if (tmax <= tmin) return 0;                                   /* :65-67 */
    }
    return 1;                                                         /* :69-78 */
}

/* ---- mesh attribute fetch (geometry.rs:223-250) ---- */
static void get_triangle_from_mesh(const orc_mesh* m, int idx, v3* a, v3* b, v3* c) {      /* :223-229 */
    uint32_t x = m->indices[idx * 3], y = m->indices[idx * 3 + 1], z = m->indices[idx * 3 + 2];
    *a = v3_from(&m->positions[x * 3]); *b = v3_from(&m->positions[y * 3]); *c = v3_from(&m->positions[z * 3]);
}
static void get_texcoords_from_mesh(const orc_mesh* m, int idx, v2* a, v2* b, v2* c) {     /* :230-236 */
    uint32_t x = m->indices[idx * 3], y = m->indices[idx * 3 + 1], z = m->indices[idx * 3 + 2];
    *a = v2_make(m->texcoords[x * 2], m->texcoords[x * 2 + 1]);
    *b = v2_make(m->texcoords[y * 2], m->texcoords[y * 2 + 1]);
    *c = v2_make(m->texcoords[z * 2], m->texcoords[z * 2 + 1]);
}
static void get_normals_from_mesh(const orc_mesh* m, int idx, v3* a, v3* b, v3* c) {       /* :237-243 */
    uint32_t x = m->indices[idx * 3], y = m->indices[idx * 3 + 1], z = m->indices[idx * 3 + 2];
    *a = v3_from(&m->normals[x * 3]); *b = v3_from(&m->normals[y * 3]); *c = v3_from(&m->normals[z * 3]);
}
/* StaticMesh::get_tangent :245-250 */
static v3 get_tangent(v2 uv1, v2 uv2, v2 uv3, v3 p1, v3 p2, v3 p3) {
    float u1 = uv1.x, u2 = uv2.x, u3 = uv3.x;
    float v1 = uv1.y, v2_ = uv2.y, v3_ = uv3.y;
    v3 num = v3_sub(v3_scale(v3_sub(p2, p1), (v3_ - v1)), v3_scale(v3_sub(p3, p1), (v2_ - v1)));
    float den = (u2 - u1) * (v3_ - v1) - (v2_ - v1) * (u3 - u1);
    return v3_divs(num, den);
}

/* IndexedTriangle::bounding_box :367-381 */
static orc_aabb indexed_triangle_bbox(const orc_mesh* m, int idx) {
    v3 a, b, c; get_triangle_from_mesh(m, idx, &a, &b, &c);
    orc_aabb r;
    r.min = v3_make(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)));
    r.max = v3_make(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)));
    return r;
}

/* impl Intersectable for IndexedTriangle :331-366.  Every candidate hit computes the
 * interpolated normal, uv and the TBN frame, as the reference does. */
static int indexed_triangle_intersect(const orc_mesh* m, int idx, const orc_ray* ray, float t_min, float t_max,
                                      orc_path* p, orc_rayhit* out) {
    if (p && p->cnt) p->cnt->tri_tests++;
    v3 a, b, c; get_triangle_from_mesh(m, idx, &a, &b, &c);           /* :333 */
    const float EPSILON = 0.0001f;                                    /* :335 */
    v3 e1 = v3_sub(b, a);                                             /* :336 */
    v3 e2 = v3_sub(c, a);                                             /* :337 */
    v3 q = v3_cross(ray->direction, e2);                              /* :338 */
    float g = v3_dot(e1, q);                                          /* :339 */
    if (fabsf(g) < EPSILON) return 0;                                 /* :340 */
    float f = 1.0f / g;                                               /* :341 */
    v3 s = v3_sub(ray->origin, a);                                    /* :342 */
    float u = f * v3_dot(s, q);                                       /* :343 */
    if (u < 0.0f) return 0;                                           /* :344 */
    v3 r = v3_cross(s, e1);                                           /* :345 */
    float v = f * v3_dot(ray->direction, r);                          /* :346 */
    if (v < 0.0f || u + v > 1.0f) return 0;                           /* :347 */
    float t = f * v3_dot(e2, r);                                      /* :348 */
    if (t < t_min || t > t_max) return 0;                             /* :349 */
    v3 na, nb, nc; get_normals_from_mesh(m, idx, &na, &nb, &nc);      /* :350 */
    /* u*nb + v*nc + (1.0-u-v)*na                                        :351 */
    v3 mesh_normal = v3_normalize(v3_add(v3_add(v3_scale(nb, u), v3_scale(nc, v)), v3_scale(na, (1.0f - u - v))));
    orc_rayhit hit = orc_rayhit_new(t, mesh_normal, orc_lambertian_default(), ray);   /* :352 */
    v2 tca, tcb, tcc; get_texcoords_from_mesh(m, idx, &tca, &tcb, &tcc);              /* :355 */
    float w = (1.0f - u - v);
    hit.has_tex_coords = 1;                                           /* :356 */
    hit.tex_coords = v2_make((u * tcb.x + v * tcc.x) + w * tca.x, (u * tcb.y + v * tcc.y) + w * tca.y);
    v3 tan_approx = get_tangent(tca, tcb, tcc, a, b, c);              /* :359 */
    v3 bitangent = v3_normalize(v3_cross(hit.normal, tan_approx));    /* :360 */
    v3 tangent = v3_normalize(v3_cross(bitangent, hit.normal));       /* :361 */
    hit.has_tangent = 1; hit.tangent = tangent;                       /* :362 */
    hit.has_bitangent = 1; hit.bitangent = bitangent;                 /* :363 */
    hit.object = idx;          /* diagnostic only */
    *out = hit;
    return 1;
}

/* ---- BVH (geometry.rs:86-123, build :175-217) ---- */

/* impl Intersectable for BVHNode :94-119 */
static int bvhnode_intersect(const orc_mesh* m, const orc_bvhnode* node, const orc_ray* ray, float t_min, float t_max,
                             orc_path* p, orc_rayhit* out) {
    if (node->has_primitive) {                                        /* :95 */
        return indexed_triangle_intersect(m, node->primitive_idx, ray, t_min, t_max, p, out);   /* :97 */
    }
    int have_best = 0;                                                /* :101 */
    orc_rayhit best_hit;
    float best_t = t_max;                                             /* :102 */
    if (orc_aabb_intersect(&node->aabb, ray, t_min, t_max, p)) {      /* :103 */
        if (node->left) {                                             /* :105 */
            orc_rayhit h;
            if (bvhnode_intersect(m, node->left, ray, t_min, t_max, p, &h)) {   /* :106 */
                best_hit = h; have_best = 1;                          /* :108 */
                best_t = h.distance;                                  /* :109 */
            }
        }
        if (node->right) {                                            /* :112 */
            orc_rayhit h;
            if (bvhnode_intersect(m, node->right, ray, t_min, best_t, p, &h)) { /* :113 */
                best_hit = h; have_best = 1;                          /* :114 */
            }
        }
    }
    if (have_best) *out = best_hit;
    return have_best;                                                 /* :117 */
}

/* StaticMesh::build_bvh_helper :190-217.  The reference sorts tris[start..end] by a
 * random axis (:200-207) but the leaf is built from `idx: start` (:194), not from the
 * sorted vector's entry, so the sort never influences the tree: the topology is the
 * index-range median split below and is independent of the RNG.  The scratch sort is
 * therefore not restated. */
static orc_bvhnode* build_bvh_helper(const orc_mesh* m, int start, int end) {
    orc_bvhnode* node = (orc_bvhnode*)calloc(1, sizeof(orc_bvhnode));    /* BVHNode::default() :191 */
    if (end - start == 1) {                                           /* :192 */
        node->aabb = indexed_triangle_bbox(m, start);                 /* :195 */
        node->has_primitive = 1; node->primitive_idx = start;         /* :194,196 */
    } else {
        int mid = start + (end - start) / 2;                          /* :209 */
        node->left = build_bvh_helper(m, start, mid);                 /* :210 */
        node->right = build_bvh_helper(m, mid, end);                  /* :211 */
        node->aabb = aabb_surrounding(&node->left->aabb, &node->right->aabb);   /* :212 */
    }
    return node;
}
void orc_mesh_build_bvh(orc_mesh* m) {                                /* :175-188 */
    if (m->bvh_root) return;
    m->bvh_root = build_bvh_helper(m, 0, m->n_triangles);
}
void orc_bvh_free(orc_bvhnode* n) {
    if (!n) return;
    orc_bvh_free(n->left); orc_bvh_free(n->right); free(n);
}

/* ---- StaticMesh (geometry.rs:253-321) ---- */

/* StaticMesh::get_material_at_uv :253-271 */
static mi_material get_material_at_uv(const orc_scene* s, const orc_mesh* m, const orc_rayhit* hit, orc_path* p) {
    if (m->material >= 0 || !hit->has_tex_coords) {                   /* :255 */
        return s->materials[m->material];                             /* :256 (unwrap) */
    }
    v2 uv = hit->tex_coords;                                          /* :259 */
    v3 albedo   = m->textures[0] >= 0 ? orc_texture_sample_v(&s->textures[m->textures[0]], uv, p) : v3_zero();   /* :260 */
    v3 emission = m->textures[1] >= 0 ? orc_texture_sample_v(&s->textures[m->textures[1]], uv, p) : v3_zero();   /* :261 */
    float metallic  = m->textures[2] >= 0 ? orc_texture_sample_v(&s->textures[m->textures[2]], uv, p).x : 0.0f;  /* :262 */
    float roughness = m->textures[3] >= 0 ? orc_texture_sample_v(&s->textures[m->textures[3]], uv, p).x : 1.0f;  /* :263 */
    mi_material r;                                                    /* :264-269 */
    r.kind = MI_MAT_PARAMETERIZED;
    r.albedo[0] = albedo.x; r.albedo[1] = albedo.y; r.albedo[2] = albedo.z;
    r.emission[0] = emission.x; r.emission[1] = emission.y; r.emission[2] = emission.z;
    r.roughness = roughness; r.metallic = metallic; r.idx_of_refraction = 0.0f;
    return r;
}

/* StaticMesh::get_adjusted_normal :274-298 */
static v3 get_adjusted_normal(const orc_scene* s, const orc_mesh* m, const orc_rayhit* hit, orc_path* p) {
    v3 n;
    if (m->textures[4] >= 0 && hit->has_tangent && hit->has_bitangent) {          /* :276-278 */
        v2 uv = hit->tex_coords;                                                  /* :280 */
        v3 smp = orc_texture_sample_v(&s->textures[m->textures[4]], uv, p);       /* :281 */
        v3 nv = v3_sub(v3_scale(smp, 2.0f), v3_make(1.0f, 1.0f, 1.0f));           /* :282 */
        m3 tbn; tbn.c0 = hit->tangent; tbn.c1 = hit->bitangent; tbn.c2 = hit->normal;
        n = m3_mul_v3(tbn, nv);                                                   /* :283 */
    } else {
        n = hit->normal;                                                          /* :286,290,294 */
    }
    return v3_normalize(m4_transpose_transform_vector(m->inv_transform, n));      /* :297 */
}

/* impl Intersectable for StaticMesh :301-314 */
int orc_mesh_intersect(const orc_scene* s, const orc_mesh* m, const orc_ray* ray, float t_min, float t_max,
                       orc_path* p, orc_rayhit* out) {
    if (p && p->cnt) p->cnt->mesh_tests++;
    if (m->bvh_root) {                                                /* :303 */
        orc_ray tr;                                                   /* :304 */
        tr.origin = m4_transform_point(m->inv_transform, ray->origin);
        tr.direction = m4_transform_vector(m->inv_transform, ray->direction);
        if (p && p->cnt && !m->bvh_root->has_primitive) {
            /* counter only: would the root box test pass?  (re-evaluated, not cached) */
            uint64_t keep = p->cnt->box_tests;
            if (orc_aabb_intersect(&m->bvh_root->aabb, &tr, t_min, t_max, p)) p->cnt->mesh_entered++;
            p->cnt->box_tests = keep;
        }
        orc_rayhit hit;
        if (bvhnode_intersect(m, m->bvh_root, &tr, t_min, t_max, p, &hit)) {     /* :305 */
            hit.hitpoint = m4_transform_point(m->transform, hit.hitpoint);        /* :307 */
            hit.normal = get_adjusted_normal(s, m, &hit, p);                      /* :308 */
            hit.material = get_material_at_uv(s, m, &hit, p);                     /* :309 */
            if (p && p->cnt) p->cnt->mesh_hits++;
            *out = hit;
            return 1;                                                             /* :310 */
        }
    }
    return 0;                                                         /* :313 */
}

/* ---- Sphere (geometry.rs:394-413) ---- */
int orc_sphere_intersect(const orc_scene* s, const mi_sphere* sp, const orc_ray* ray, float t_min, float t_max, orc_rayhit* out) {
    v3 center = v3_from(sp->center);
    v3 f = v3_sub(ray->origin, center);                               /* :397 */
    float a = v3_mag2(ray->direction);                                /* :398 */
    float b = 2.0f * v3_dot(f, ray->direction);                       /* :399 */
    float c = v3_mag2(f) - sp->radius * sp->radius;                   /* :400 */
    float d = b * b - 4.0f * a * c;                                   /* :401 */
    if (d < 0.0f) return 0;                                           /* :402 */
    float t1 = (-b - sqrtf(d)) / (2.0f * a);                          /* :406 */
    float t2 = (-b + sqrtf(d)) / (2.0f * a);                          /* :407 */
    float t = (t1 >= t_min) ? t1 : t2;                                /* :408 */
    v3 hitpoint = v3_add(ray->origin, v3_scale(ray->direction, t));   /* :409 */
    if (t < t_min || t > t_max) return 0;                             /* :410 */
    mi_material mat = (s && sp->material >= 0) ? s->materials[sp->material] : orc_lambertian_default();
    *out = orc_rayhit_new(t, v3_normalize(v3_sub(hitpoint, center)), mat, ray);   /* :411 */
    return 1;
}

/* ---- Triangle (geometry.rs:430-450) ---- */
int orc_triangle_intersect(const orc_scene* s, const mi_triangle* tr, const orc_ray* ray, float t_min, float t_max, orc_rayhit* out) {
    const float EPSILON = 0.0001f;                                    /* :433 */
    v3 A = v3_from(tr->a), B = v3_from(tr->b), C = v3_from(tr->c);
    v3 e1 = v3_sub(B, A);                                             /* :434 */
    v3 e2 = v3_sub(C, A);                                             /* :435 */
    v3 q = v3_cross(ray->direction, e2);                              /* :436 */
    float a = v3_dot(e1, q);                                          /* :437 */
    if (fabsf(a) < EPSILON) return 0;                                 /* :438 */
    float f = 1.0f / a;                                               /* :439 */
    v3 sv = v3_sub(ray->origin, A);                                   /* :440 */
    float u = f * v3_dot(sv, q);                                      /* :441 */
    if (u < 0.0f) return 0;                                           /* :442 */
    v3 r = v3_cross(sv, e1);                                          /* :443 */
    float v = f * v3_dot(ray->direction, r);                          /* :444 */
    if (v < 0.0f || u + v > 1.0f) return 0;                           /* :445 */
    float t = f * v3_dot(e2, r);                                      /* :446 */
    if (t < t_min || t > t_max) return 0;                             /* :447 */
    *out = orc_rayhit_new(t, v3_normalize(v3_cross(e1, e2)), s->materials[tr->material], ray);   /* :449 */
    return 1;
}

/* ---- Plane (geometry.rs:473-489) ---- */
int orc_plane_intersect(const orc_scene* s, const mi_plane* pl, const orc_ray* ray, float t_min, float t_max, orc_rayhit* out) {
    v3 normal = v3_from(pl->normal);
    v3 to_ray_origin = v3_sub(ray->origin, v3_from(pl->point));       /* :476 */
    float origin_dist = v3_dot(to_ray_origin, normal);                /* :477 */
    v3 n = v3_scale(normal, orc_signum(origin_dist));                 /* :478 */
    float d = v3_dot(ray->direction, n);                              /* :479 */
    if (d >= 0.0f) return 0;                                          /* :480 */
    float t = fabsf(origin_dist) / fabsf(d);                          /* :484 */
    if (t < t_min || t > t_max) return 0;                             /* :485 */
    *out = orc_rayhit_new(t, n, s->materials[pl->material], ray);     /* :487 */
    return 1;
}

/* ---- ConvexVolume (geometry.rs:501-526) ---- */
/* `self.boundary.intersect_ray(ray, t_min, t_max)` (:505,508) for the boundary kinds the ABI carries: dyn dispatch over
 * Sphere (inline), Triangle, Plane, StaticMesh, or a nested Scene (tracing.rs:327-346: closest hit, strict `<`). */
static int boundary_entry_intersect(const orc_scene* s, int kind, int index, const orc_ray* ray, float t_min, float t_max,
                                    orc_path* p, orc_rayhit* out) {
    switch (kind) {
    case MI_OBJ_SPHERE:   return orc_sphere_intersect(s, &s->spheres[index], ray, t_min, t_max, out);
    case MI_OBJ_TRIANGLE: return orc_triangle_intersect(s, &s->triangles[index], ray, t_min, t_max, out);
    case MI_OBJ_PLANE:    return orc_plane_intersect(s, &s->planes[index], ray, t_min, t_max, out);
    case MI_OBJ_MESH:     return orc_mesh_intersect(s, &s->meshes[index], ray, t_min, t_max, p, out);
    default: return 0;
    }
}
static int boundary_intersect(const orc_scene* s, const mi_volume* vo, const orc_ray* ray, float t_min, float t_max,
                              orc_path* p, orc_rayhit* out) {
    if (vo->boundary_kind == MI_OBJ_SPHERE) {
        mi_sphere boundary;
        boundary.center[0] = vo->boundary_center[0]; boundary.center[1] = vo->boundary_center[1]; boundary.center[2] = vo->boundary_center[2];
        boundary.radius = vo->boundary_radius; boundary.material = -1;   /* boundary material is never read (:503 "arbitrary") */
        return orc_sphere_intersect(NULL, &boundary, ray, t_min, t_max, out);
    }
    /* the counters describe Scene.objects' own work: a boundary's mesh walk is not counted */
    orc_path q = *p; q.cnt = NULL;
    if (vo->boundary_kind == MI_OBJ_SCENE) {                            /* impl Intersectable for Scene, tracing.rs:327-346 */
        int have_best = 0; orc_rayhit best_hit;
        for (int k = 0; k < vo->boundary_count; k++) {
            const mi_object* e = &s->boundary_objects[vo->boundary_index + k];
            orc_rayhit hit;
            if (boundary_entry_intersect(s, e->kind, e->index, ray, t_min, t_max, &q, &hit)) {
                if (!have_best) { best_hit = hit; have_best = 1; }      /* :333 */
                else if (hit.distance < best_hit.distance) best_hit = hit;   /* :335-336 */
            }
        }
        if (have_best) *out = best_hit;
        return have_best;
    }
    return boundary_entry_intersect(s, vo->boundary_kind, vo->boundary_index, ray, t_min, t_max, &q, out);
}

int orc_volume_intersect(const orc_scene* s, const mi_volume* vo, const orc_ray* ray, float t_min, float t_max,
                         orc_path* p, orc_rayhit* out) {
    const float F32_MIN = -3.40282347e+38f, F32_MAX = 3.40282347e+38f;
    orc_rayhit hit_entr, hit_exit;
    if (!boundary_intersect(s, vo, ray, F32_MIN, F32_MAX, p, &hit_entr)) return 0;                 /* :505-506 */
    float t_entr = hit_entr.distance;                                                             /* :507 */
    if (!boundary_intersect(s, vo, ray, t_entr + 0.0001f, F32_MAX, p, &hit_exit)) return 0;        /* :508-509 */
    float t_exit = hit_exit.distance;                                 /* :510 */
    if (t_exit < t_min || t_entr > t_max) return 0;                   /* :512 */
    float t_start = fmaxf(t_entr, t_min);                             /* :513 */
    float t_end = fminf(t_exit, t_max);                               /* :514 */
    float dist_in_volume = t_end - t_start;                           /* :515 */
    float dist_before_scatter = (-1.0f / vo->density) * orc_logf(orc_gen_range_01(&p->rng));    /* :517 */
    if (dist_before_scatter < dist_in_volume) {                       /* :518 */
        *out = orc_rayhit_new(t_start + dist_before_scatter, v3_zero(), s->materials[vo->phase_material], ray);   /* :520 */
        return 1;
    }
    return 0;                                                         /* :524 */
}
