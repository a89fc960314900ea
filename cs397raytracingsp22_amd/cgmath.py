"""Scene-authoring helpers with cgmath's conventions (column-major Matrix4, M*v).

Only what the reference's scene literal uses (tracing.rs:383,393,403):
Matrix4::from_translation / from_angle_x / from_angle_y / from_angle_z / from_scale and
Matrix4 products, plus the inverse that StaticMesh::load_from_file stores
(geometry.rs:168).  Matrices are numpy float32 4x4 arrays indexed [row, col]; `cols16`
gives the column-major float[16] the C ABI carries (mi_mesh.transform).
"""
from __future__ import annotations

import math
import numpy as np


def vec3(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def identity():
    return np.eye(4, dtype=np.float32)


def from_translation(v):
    m = identity()
    m[0:3, 3] = np.asarray(v, dtype=np.float32)
    return m


def from_scale(s):
    m = identity()
    m[0, 0] = m[1, 1] = m[2, 2] = np.float32(s)
    return m


def from_nonuniform_scale(x, y, z):
    m = identity()
    m[0, 0], m[1, 1], m[2, 2] = np.float32(x), np.float32(y), np.float32(z)
    return m


def _sc(deg):
    r = np.float32(math.radians(deg))
    return np.float32(math.sin(r)), np.float32(math.cos(r))


def from_angle_x(deg):
    s, c = _sc(deg)
    m = identity()
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def from_angle_y(deg):
    s, c = _sc(deg)
    m = identity()
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def from_angle_z(deg):
    s, c = _sc(deg)
    m = identity()
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def mul(*ms):
    out = ms[0].astype(np.float32)
    for m in ms[1:]:
        out = (out @ m.astype(np.float32)).astype(np.float32)
    return out


def inverse_transform(m):
    """Matrix4::inverse_transform (general inverse); computed in f64, rounded to f32."""
    return np.linalg.inv(m.astype(np.float64)).astype(np.float32)


def cols16(m):
    """column-major float[16] (cgmath memory order)"""
    return np.ascontiguousarray(m.astype(np.float32).T).reshape(16)
