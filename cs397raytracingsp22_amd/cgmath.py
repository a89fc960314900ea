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
    # cgmath 0.18 `impl From<Deg<f32>> for Rad<f32>`: deg * (PI / 180 as f32) — ONE f32 product of the f32-rounded constant (not the f64
    # product rounded once: the two differ in the last bit for ~9 % of angles; equal for every angle run() uses: 45, -60, 180, -90)
    r = np.float32(deg) * np.float32(math.pi / 180.0)
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
    """Matrix4 * Matrix4 with cgmath's arithmetic: column c of the product is self * rhs[c], and Matrix4 * Vector4 is
    ((col0 * x + col1 * y) + col2 * z) + col3 * w — every operation a separate f32 rounding (Rust never fuses).  Written out
    (not numpy's matmul, whose summation order and FMA use belong to the BLAS underneath) so that the compiled mirror
    (host/geometry.hpp Matrix4::operator*) produces the same bits."""
    out = ms[0].astype(np.float32)
    for m in ms[1:]:
        r = m.astype(np.float32)
        prod = np.empty((4, 4), np.float32)
        for c in range(4):
            x, y, z, w = r[0, c], r[1, c], r[2, c], r[3, c]
            prod[:, c] = ((out[:, 0] * x + out[:, 1] * y) + out[:, 2] * z) + out[:, 3] * w
        out = prod
    return out


def inverse_transform(m):
    """Matrix4::inverse_transform (geometry.rs:168): the general inverse.  Gauss-Jordan with partial pivoting in f64, rounded
    to f32 at the end — one fixed sequence of IEEE double operations, the same the compiled mirror runs
    (host/geometry.hpp Matrix4::inverse_transform), so both mirrors hand the library the same 16 floats.  cgmath itself
    inverts with f32 cofactors; its exact operation order is not restated here (parity unpinned)."""
    a = [[float(m[r, c]) for c in range(4)] + [1.0 if r == c else 0.0 for c in range(4)] for r in range(4)]
    for i in range(4):
        p = i
        for r in range(i + 1, 4):
            if abs(a[r][i]) > abs(a[p][i]):
                p = r
        if abs(a[p][i]) < 1e-30:
            raise np.linalg.LinAlgError("transform is not invertible")
        a[i], a[p] = a[p], a[i]
        d = a[i][i]
        a[i] = [v / d for v in a[i]]
        for r in range(4):
            if r != i:
                f = a[r][i]
                a[r] = [a[r][c] - f * a[i][c] for c in range(8)]
    return np.array([[a[r][4 + c] for c in range(4)] for r in range(4)], dtype=np.float64).astype(np.float32)


def cols16(m):
    """column-major float[16] (cgmath memory order)"""
    return np.ascontiguousarray(m.astype(np.float32).T).reshape(16)
