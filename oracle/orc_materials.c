/*
 * orc_materials.c — TEST INFRASTRUCTURE (oracle).  Restates src/util/materials.rs of the
 * reference: Lambertian, Metal, Dielectric, ParameterizedMaterial, Isotropic and
 * sample_hemisphere.  Citations are materials.rs unless noted.  alpha_sample (:181-193)
 * and rtow_sample (:196-199) are dead code in the reference and are not restated.
 */
#include "orc_internal.h"

static const float ORC_PI = 3.14159265358979323846f;    /* std::f32::consts::PI */

mi_material orc_lambertian_default(void) {               /* impl Default for Lambertian :24-31 */
    mi_material m;
    m.kind = MI_MAT_LAMBERTIAN;
    m.albedo[0] = m.albedo[1] = m.albedo[2] = 1.0f;
    m.emission[0] = m.emission[1] = m.emission[2] = 0.0f;
    m.roughness = 0.0f; m.metallic = 0.0f; m.idx_of_refraction = 0.0f;
    return m;
}

/* sample_hemisphere :171-178 */
static void sample_hemisphere(const orc_rayhit* hit, orc_path* p, v3* dir_out, float* pdf_out) {
    v3 dir = orc_rand_sphere_vec(p);                                  /* :173 */
    dir.y = fabsf(dir.y);                                             /* :174 */
    m3 rotation = orc_between_vectors(v3_make(0.0f, 1.0f, 0.0f), hit->normal);   /* :176 */
    *dir_out = m3_mul_v3(rotation, dir);                              /* :177 rotate_vector */
    *pdf_out = 1.0f / (2.0f * ORC_PI);
}

void orc_material_scatter(const mi_material* m, const orc_rayhit* hit, const orc_ray* ray, orc_path* p,
                          orc_ray* new_ray, v3* brdf, float* pdf) {
    v3 albedo = v3_from(m->albedo);
    switch (m->kind) {
    case MI_MAT_LAMBERTIAN: {                                         /* :34-44 */
        v3 dir; float pd;
        sample_hemisphere(hit, p, &dir, &pd);                         /* :35 */
        new_ray->origin = hit->hitpoint; new_ray->direction = dir;    /* :37-40 */
        *brdf = v3_divs(albedo, ORC_PI);                              /* :41 */
        *pdf = pd;                                                    /* :42 */
        break;
    }
    case MI_MAT_METAL: {                                              /* :57-66 */
        v3 refl = orc_reflect_v(ray->direction, hit->normal);
        v3 fuzz = v3_scale(orc_rand_sphere_vec(p), m->roughness);
        new_ray->origin = hit->hitpoint;
        new_ray->direction = v3_add(refl, fuzz);                      /* :62 */
        *brdf = albedo;                                               /* :64 */
        *pdf = 1.0f;                                                  /* :65 */
        break;
    }
    case MI_MAT_DIELECTRIC: {                                         /* :78-99 */
        float ior = m->idx_of_refraction;
        float eta = hit->frontface ? 1.0f / ior : ior;                /* :80 */
        float cosv = fminf(-v3_dot(ray->direction, hit->normal), 1.0f);
        int critical_angle = eta * sqrtf(1.0f - orc_powi2(cosv)) > 1.0f;          /* :81 */
        float fresnel_factor = orc_fresnel_v(ray->direction, hit->normal, ior);  /* :82 */
        /* :84 — `&&` short-circuits: the RNG is drawn only when !critical_angle */
        int will_refract = !critical_angle && (orc_gen_range_01(&p->rng) >= fresnel_factor);
        v3 new_dir = will_refract ? orc_refract_v(ray->direction, hit->normal, eta)   /* :86 */
                                  : orc_reflect_v(ray->direction, hit->normal);       /* :89 */
        new_ray->origin = hit->hitpoint; new_ray->direction = new_dir;
        *brdf = v3_make(1.0f, 1.0f, 1.0f);                            /* :97 */
        *pdf = 1.0f;                                                  /* :98 */
        break;
    }
    case MI_MAT_PARAMETERIZED: {                                      /* :114-143 */
        float fresnel = orc_fresnel_v(ray->direction, hit->normal, 1.5f);         /* :116 */
        float k_s = fresnel * (1.0f - m->roughness);                  /* :117 */
        float k_d = (1.0f - k_s) * (1.0f - m->metallic);              /* :118 */
        if (orc_gen_range_01(&p->rng) < k_d) {                        /* :120 */
            v3 dir; float pd;
            sample_hemisphere(hit, p, &dir, &pd);                     /* :122 */
            new_ray->origin = hit->hitpoint; new_ray->direction = dir;
            *brdf = v3_divs(albedo, ORC_PI);                          /* :128 */
            *pdf = pd;
        } else {
            v3 refl = orc_reflect_v(ray->direction, hit->normal);
            v3 fuzz = v3_scale(orc_rand_sphere_vec(p), m->roughness);
            new_ray->origin = hit->hitpoint;
            new_ray->direction = v3_add(refl, fuzz);                  /* :137 */
            /* lerpvec(1, albedo, metallic) = (1-k)*a + k*b, tracing.rs:95-97    :139 */
            float k = m->metallic;
            *brdf = v3_add(v3_scale(v3_make(1.0f, 1.0f, 1.0f), (1.0f - k)), v3_scale(albedo, k));
            *pdf = 1.0f;
        }
        break;
    }
    case MI_MAT_ISOTROPIC:                                            /* :159-162 */
    default:
        new_ray->origin = hit->hitpoint;
        new_ray->direction = orc_rand_sphere_vec(p);
        *brdf = albedo;
        *pdf = 1.0f;
        break;
    }
}

/* Material::emission :45-47, 68-70, 101-103, 146-148, 163-165 */
v3 orc_material_emission(const mi_material* m) {
    if (m->kind == MI_MAT_DIELECTRIC) return v3_zero();               /* :102 */
    return v3_from(m->emission);
}
