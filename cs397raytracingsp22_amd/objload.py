"""OBJ reader with the behaviour the reference gets from tobj 3.2.0
`load_obj(.., LoadOptions{single_index: true, triangulate: true, ..})`
(geometry.rs:140-148).  Load-time work, outside the accelerated path; it exists so
that StaticMesh.load_from_file keeps the reference's signature.

tobj semantics restated (crate source not in /root/reference — restated from its
documented behaviour):
  * `v`, `vt` (first two components), `vn`, `f` records; 1-based and negative indices
  * polygons are fan-triangulated (0, k, k+1) in file order
  * single_index: each distinct (v, vt, vn) tuple becomes one vertex, numbered by first use
  * an `o` or `g` record closes the current model when it already holds faces;
    the reference keeps only models[0] (geometry.rs:157)
"""
from __future__ import annotations

import numpy as np


class Mesh:
    """tobj::Mesh: positions (3n), normals (3n), texcoords (2n), indices (3t)."""

    def __init__(self, positions, normals, texcoords, indices, name=""):
        self.positions = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1)
        self.normals = np.ascontiguousarray(normals, dtype=np.float32).reshape(-1)
        self.texcoords = np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1)
        self.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        self.name = name

    @property
    def n_vertices(self):
        return self.positions.size // 3

    @property
    def n_triangles(self):
        return self.indices.size // 3


def _fix(idx, n):
    return idx - 1 if idx > 0 else n + idx


def load_obj_text(text: str):
    pos, tex, nrm = [], [], []
    models = []
    faces = []  # list of list of (v, vt, vn)
    name = "unnamed_object"

    def flush():
        nonlocal faces
        if not faces:
            return
        index_map = {}
        P, N, T, I = [], [], [], []
        for poly in faces:
            tris = [(poly[0], poly[k], poly[k + 1]) for k in range(1, len(poly) - 1)]
            for tri in tris:
                for key in tri:
                    i = index_map.get(key)
                    if i is None:
                        i = len(index_map)
                        index_map[key] = i
                        v, vt, vn = key
                        P.extend(pos[v])
                        if tex and vt is not None:
                            T.extend(tex[vt])
                        if nrm and vn is not None:
                            N.extend(nrm[vn])
                    I.append(i)
        models.append(Mesh(P, N, T, I, name))
        faces = []

    for line in text.splitlines():
        parts = line.split()
        if not parts or parts[0].startswith("#"):
            continue
        tag = parts[0]
        if tag == "v":
            pos.append((float(parts[1]), float(parts[2]), float(parts[3])))
        elif tag == "vt":
            tex.append((float(parts[1]), float(parts[2]) if len(parts) > 2 else 0.0))
        elif tag == "vn":
            nrm.append((float(parts[1]), float(parts[2]), float(parts[3])))
        elif tag == "f":
            poly = []
            for tok in parts[1:]:
                f = tok.split("/")
                v = _fix(int(f[0]), len(pos))
                vt = _fix(int(f[1]), len(tex)) if len(f) > 1 and f[1] else None
                vn = _fix(int(f[2]), len(nrm)) if len(f) > 2 and f[2] else None
                poly.append((v, vt, vn))
            if len(poly) >= 3:
                faces.append(poly)
        elif tag in ("o", "g"):
            flush()
            name = " ".join(parts[1:]) if len(parts) > 1 else name
    flush()
    return models


def load_obj(path: str):
    with open(path, "r") as fh:
        return load_obj_text(fh.read())
