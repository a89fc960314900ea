// materials.hpp — C++ host mirror of src/util/materials.rs: `trait Material` and its five
// implementors with the reference's names and fields.  scatter()/emission() run on the GPU;
// these types only carry parameters and flatten themselves into mi_material.
#pragma once
#include <array>
#include <memory>
#include "../../include/mi_rt.h"

namespace cs397 {

using Vec3 = std::array<float, 3>;     // tracing.rs:22 `type Vec3 = Vector3<f32>`
using Color = Vec3;                    // tracing.rs:24

// trait Material (materials.rs:12-15) + the additive `flatten`
struct Material {
    virtual ~Material() = default;
    virtual mi_material flatten() const = 0;
};
using MaterialRef = std::shared_ptr<const Material>;     // Arc<dyn Material + Send + Sync>

namespace detail {
inline mi_material pod(int kind, Vec3 albedo, Vec3 emission, float rough, float metal, float ior) {
    mi_material m{};
    m.kind = kind;
    for (int i = 0; i < 3; i++) { m.albedo[i] = albedo[i]; m.emission[i] = emission[i]; }
    m.roughness = rough; m.metallic = metal; m.idx_of_refraction = ior;
    return m;
}
}  // namespace detail

struct Lambertian : Material {                         // materials.rs:20-23; Default :24-31
    Color albedo{1.0f, 1.0f, 1.0f};
    Color emission{0.0f, 0.0f, 0.0f};
    Lambertian() = default;
    Lambertian(Color a, Color e = {0, 0, 0}) : albedo(a), emission(e) {}
    mi_material flatten() const override { return detail::pod(MI_MAT_LAMBERTIAN, albedo, emission, 0, 0, 0); }
};
struct Metal : Material {                              // materials.rs:51-55
    Color albedo{1, 1, 1}; Color emission{0, 0, 0}; float roughness = 0.0f;
    Metal(Color a, Color e, float r) : albedo(a), emission(e), roughness(r) {}
    mi_material flatten() const override { return detail::pod(MI_MAT_METAL, albedo, emission, roughness, 0, 0); }
};
struct Dielectric : Material {                         // materials.rs:74-76
    float idx_of_refraction = 1.5f;
    explicit Dielectric(float ior) : idx_of_refraction(ior) {}
    mi_material flatten() const override { return detail::pod(MI_MAT_DIELECTRIC, {0, 0, 0}, {0, 0, 0}, 0, 0, idx_of_refraction); }
};
struct ParameterizedMaterial : Material {              // materials.rs:107-112
    Color albedo{1, 1, 1}; Color emission{0, 0, 0}; float roughness = 1.0f; float metallic = 0.0f;
    ParameterizedMaterial(Color a, Color e, float r, float m) : albedo(a), emission(e), roughness(r), metallic(m) {}
    mi_material flatten() const override { return detail::pod(MI_MAT_PARAMETERIZED, albedo, emission, roughness, metallic, 0); }
};
struct Isotropic : Material {                          // materials.rs:152-157
    Color albedo{1, 1, 1}; Color emission{0, 0, 0};
    Isotropic(Color a, Color e = {0, 0, 0}) : albedo(a), emission(e) {}
    mi_material flatten() const override { return detail::pod(MI_MAT_ISOTROPIC, albedo, emission, 0, 0, 0); }
};

}  // namespace cs397
