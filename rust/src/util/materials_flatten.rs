// materials_flatten.rs — pasted into src/util/materials.rs by `include!("materials_flatten.rs");` (last line of that file).
// The five `Material` implementors (materials.rs:20,51,74,107,152) as mi_material PODs.  Fields the kind does not have are 0:
// the kernels read only what the reference's scatter() / emission() of that kind read.
// Needs: `pub trait Material: super::mi_rt::FlattenMaterial` (materials.rs:12).  UNVERIFIED by a compiler (see mi_rt.rs).

use super::mi_rt::{mi_material, FlattenMaterial,
                   MI_MAT_DIELECTRIC, MI_MAT_ISOTROPIC, MI_MAT_LAMBERTIAN, MI_MAT_METAL, MI_MAT_PARAMETERIZED};

impl FlattenMaterial for Lambertian {                    // materials.rs:20-23
    fn flatten(&self) -> mi_material {
        mi_material { kind: MI_MAT_LAMBERTIAN, albedo: self.albedo.into(), emission: self.emission.into(),
                      roughness: 0.0, metallic: 0.0, idx_of_refraction: 0.0 }
    }
}
impl FlattenMaterial for Metal {                         // materials.rs:51-55
    fn flatten(&self) -> mi_material {
        mi_material { kind: MI_MAT_METAL, albedo: self.albedo.into(), emission: self.emission.into(),
                      roughness: self.roughness, metallic: 0.0, idx_of_refraction: 0.0 }
    }
}
impl FlattenMaterial for Dielectric {                    // materials.rs:74-76; emission() is zero (materials.rs:101-103)
    fn flatten(&self) -> mi_material {
        mi_material { kind: MI_MAT_DIELECTRIC, albedo: [0.0, 0.0, 0.0], emission: [0.0, 0.0, 0.0],
                      roughness: 0.0, metallic: 0.0, idx_of_refraction: self.idx_of_refraction }
    }
}
impl FlattenMaterial for ParameterizedMaterial {         // materials.rs:107-112
    fn flatten(&self) -> mi_material {
        mi_material { kind: MI_MAT_PARAMETERIZED, albedo: self.albedo.into(), emission: self.emission.into(),
                      roughness: self.roughness, metallic: self.metallic, idx_of_refraction: 0.0 }
    }
}
impl FlattenMaterial for Isotropic {                     // materials.rs:152-157
    fn flatten(&self) -> mi_material {
        mi_material { kind: MI_MAT_ISOTROPIC, albedo: self.albedo.into(), emission: self.emission.into(),
                      roughness: 0.0, metallic: 0.0, idx_of_refraction: 0.0 }
    }
}
