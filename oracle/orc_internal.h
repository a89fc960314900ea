/* orc_internal.h — TEST INFRASTRUCTURE (oracle); shared types of the oracle's C files. */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "orc.h"
#include "orc_math.h"
#include "orc_rng.h"

/* Ray (tracing.rs:104-107) */
typedef struct { v3 origin, direction; } orc_ray;

/* RayHit (tracing.rs:109-118).  `material` is an Arc<dyn Material> in the reference;
 * here the resolved material value (StaticMesh synthesises one per hit, geometry.rs:264). */
typedef struct {
    float distance;
    v3    hitpoint;
    v3    normal;
    mi_material material;
    int   frontface;
    int   has_tex_coords;  v2 tex_coords;
    int   has_tangent;     v3 tangent;
    int   has_bitangent;   v3 bitangent;
    int   object;          /* diagnostic: index into Scene.objects (-1 inside a mesh) */
} orc_rayhit;

/* AABB (geometry.rs:22-25) */
typedef struct { v3 min, max; } orc_aabb;

struct orc_mesh;

/* BVHNode (geometry.rs:87-92): Option<Box<..>> children, Option<IndexedTriangle> primitive */
typedef struct orc_bvhnode {
    orc_aabb aabb;
    struct orc_bvhnode* left;
    struct orc_bvhnode* right;
    int has_primitive;
    int primitive_idx;     /* IndexedTriangle.idx (geometry.rs:327) */
} orc_bvhnode;

/* StaticMesh (geometry.rs:127-134) with tobj's Mesh arrays */
typedef struct orc_mesh {
    float*    positions; float* normals; float* texcoords; uint32_t* indices;
    int       n_vertices, n_triangles;
    float     transform[16], inv_transform[16];
    int       material;            /* -1 = None */
    int       textures[5];         /* -1 = None */
    orc_bvhnode* bvh_root;
} orc_mesh;

struct orc_scene {
    float point_light_pos[3], ambient[3];              /* tracing.rs:216-217 (Phong only) */
    mi_object*   objects;   int n_objects;
    mi_sphere*   spheres;   int n_spheres;
    mi_triangle* triangles; int n_triangles;
    mi_plane*    planes;    int n_planes;
    mi_volume*   volumes;   int n_volumes;
    orc_mesh*    meshes;    int n_meshes;
    mi_material* materials; int n_materials;
    mi_texture*  textures;  int n_textures;   /* rgb deep-copied */
    mi_object*   boundary_objects; int n_boundary_objects;   /* entries of nested Scenes used as ConvexVolume boundaries */
};

/* per-path context: RNG stream, signature accumulator, counters */
typedef struct {
    orc_rng rng;
    uint32_t sig;
    orc_counters* cnt;
} orc_path;

static inline v3 v3_from(const float* p) { return v3_make(p[0], p[1], p[2]); }

/* tracing.rs */
v3    orc_reflect_v(v3 v, v3 n);
float orc_fresnel_v(v3 v, v3 n, float ir);
v3    orc_refract_v(v3 v, v3 n, float eta);
v3    orc_rand_sphere_vec(orc_path* p);
v3    orc_rand_disk_vec(orc_path* p);
orc_rayhit orc_rayhit_new(float distance, v3 normal, mi_material material, const orc_ray* ray);
int   orc_scene_intersect_ray(const orc_scene* s, const orc_ray* ray, float t_min, float t_max, orc_path* p, orc_rayhit* out);

/* geometry.rs */
int orc_sphere_intersect(const orc_scene* s, const mi_sphere* sp, const orc_ray* ray, float t_min, float t_max, orc_rayhit* out);
int orc_triangle_intersect(const orc_scene* s, const mi_triangle* tr, const orc_ray* ray, float t_min, float t_max, orc_rayhit* out);
int orc_plane_intersect(const orc_scene* s, const mi_plane* pl, const orc_ray* ray, float t_min, float t_max, orc_rayhit* out);
int orc_volume_intersect(const orc_scene* s, const mi_volume* vo, const orc_ray* ray, float t_min, float t_max, orc_path* p, orc_rayhit* out);
int orc_mesh_intersect(const orc_scene* s, const orc_mesh* m, const orc_ray* ray, float t_min, float t_max, orc_path* p, orc_rayhit* out);
int orc_aabb_intersect(const orc_aabb* b, const orc_ray* ray, float t_min, float t_max, orc_path* p);
void orc_mesh_build_bvh(orc_mesh* m);
void orc_bvh_free(orc_bvhnode* n);

/* materials.rs */
void orc_material_scatter(const mi_material* m, const orc_rayhit* hit, const orc_ray* ray, orc_path* p,
                          orc_ray* new_ray, v3* brdf, float* pdf);
v3   orc_material_emission(const mi_material* m);
mi_material orc_lambertian_default(void);

/* texture.rs */
v3 orc_texture_sample_v(const mi_texture* t, v2 uv, orc_path* p);

#endif
