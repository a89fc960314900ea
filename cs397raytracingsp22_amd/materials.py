"""Host-side mirror of src/util/materials.rs: the five `Material` implementors as plain
value types (same names and fields).  `scatter`/`emission` themselves run on the GPU
(csrc/pt_kernels.hip); these classes only carry parameters across the C ABI."""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np

from . import abi


def _v(x):
    return np.asarray(x, dtype=np.float32).reshape(3)


_ZERO = (0.0, 0.0, 0.0)


class Material:
    """trait Material (materials.rs:12-15)"""
    kind = -1

    def to_pod(self) -> abi.mi_material:
        m = abi.mi_material()
        m.kind = self.kind
        m.albedo = abi.f3(*_v(getattr(self, "albedo", _ZERO)))
        m.emission = abi.f3(*_v(getattr(self, "emission", _ZERO)))
        m.roughness = float(getattr(self, "roughness", 0.0))
        m.metallic = float(getattr(self, "metallic", 0.0))
        m.idx_of_refraction = float(getattr(self, "idx_of_refraction", 0.0))
        return m


@dataclass
class Lambertian(Material):          # materials.rs:20-23, Default :24-31
    albedo: tuple = (1.0, 1.0, 1.0)
    emission: tuple = _ZERO
    kind = abi.MI_MAT_LAMBERTIAN


@dataclass
class Metal(Material):               # materials.rs:51-55
    albedo: tuple = (1.0, 1.0, 1.0)
    emission: tuple = _ZERO
    roughness: float = 0.0
    kind = abi.MI_MAT_METAL


@dataclass
class Dielectric(Material):          # materials.rs:74-76
    idx_of_refraction: float = 1.5
    kind = abi.MI_MAT_DIELECTRIC


@dataclass
class ParameterizedMaterial(Material):   # materials.rs:107-112
    albedo: tuple = (1.0, 1.0, 1.0)
    emission: tuple = _ZERO
    roughness: float = 1.0
    metallic: float = 0.0
    kind = abi.MI_MAT_PARAMETERIZED


@dataclass
class Isotropic(Material):           # materials.rs:152-157
    albedo: tuple = (1.0, 1.0, 1.0)
    emission: tuple = _ZERO
    kind = abi.MI_MAT_ISOTROPIC
