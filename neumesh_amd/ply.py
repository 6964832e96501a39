"""Minimal PLY reader/writer for the prior mesh (the reference reads it with
``open3d.io.read_triangle_mesh``, models/frameworks/neumesh/__init__.py:14; open3d is not a
dependency here).  Supports ascii and binary_little_endian, vertex x/y/z (+ optional nx/ny/nz),
triangular faces; vertex normals are computed area-weighted when the file has none."""
from __future__ import annotations

import numpy as np

_PLY_DTYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4",
               "float": "f4", "double": "f8", "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2",
               "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}


class TriangleMesh:
    """Duck-type of the open3d mesh attributes NeuMesh touches (models/mesh_grid.py:19-24,60-63)."""

    def __init__(self, vertices, triangles=None, vertex_normals=None):
        self.vertices = np.asarray(vertices, dtype=np.float64).reshape(-1, 3)
        self.triangles = np.zeros((0, 3), np.int64) if triangles is None else np.asarray(triangles, np.int64).reshape(-1, 3)
        self.vertex_normals = None if vertex_normals is None else np.asarray(vertex_normals, np.float64).reshape(-1, 3)

    def compute_vertex_normals(self):
        if self.vertex_normals is not None and len(self.vertex_normals) == len(self.vertices):
            return self
        n = np.zeros_like(self.vertices)
        if len(self.triangles):
            a, b, c = (self.vertices[self.triangles[:, i]] for i in range(3))
            fn = np.cross(b - a, c - a)  # length = 2*area: area-weighted accumulation
            for i in range(3):
                np.add.at(n, self.triangles[:, i], fn)
        norm = np.linalg.norm(n, axis=1, keepdims=True)
        self.vertex_normals = np.where(norm > 0, n / np.maximum(norm, 1e-30), np.array([0.0, 0.0, 1.0]))
        return self


def read_ply(path: str) -> TriangleMesh:
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                elements[-1]["props"].append(tok[1:])
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian"):
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        verts = normals = faces = None
        for el in elements:
            scalar = all(p[0] != "list" for p in el["props"])
            if scalar:
                names = [p[1] for p in el["props"]]
                if fmt == "ascii":
                    data = np.loadtxt([f.readline().decode() for _ in range(el["count"])], ndmin=2) if el["count"] else np.zeros((0, len(names)))
                    cols = {n: data[:, i] for i, n in enumerate(names)}
                else:
                    dt = np.dtype([(p[1], "<" + _PLY_DTYPES[p[0]]) for p in el["props"]])
                    rec = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt)
                    cols = {n: rec[n] for n in names}
                if el["name"] == "vertex":
                    verts = np.stack([cols["x"], cols["y"], cols["z"]], -1).astype(np.float64)
                    if all(k in cols for k in ("nx", "ny", "nz")):
                        normals = np.stack([cols["nx"], cols["ny"], cols["nz"]], -1).astype(np.float64)
            else:
                rows = []
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        t = f.readline().split()
                        rows.append([int(x) for x in t[1:1 + int(t[0])]])
                else:
                    p = el["props"][0]
                    ct, it = np.dtype("<" + _PLY_DTYPES[p[1]]), np.dtype("<" + _PLY_DTYPES[p[2]])
                    if len(el["props"]) != 1:
                        raise ValueError(f"{path}: mixed list/scalar face properties unsupported")
                    for _ in range(el["count"]):
                        k = int(np.frombuffer(f.read(ct.itemsize), ct)[0])
                        rows.append(np.frombuffer(f.read(it.itemsize * k), it).astype(np.int64).tolist())
                if el["name"] == "face":
                    tris = []
                    for r in rows:  # fan-triangulate polygons
                        tris.extend([r[0], r[i], r[i + 1]] for i in range(1, len(r) - 1))
                    faces = np.asarray(tris, np.int64).reshape(-1, 3)
        if verts is None:
            raise ValueError(f"{path}: no vertex element")
        return TriangleMesh(verts, faces, normals)


def write_ply(path: str, vertices, triangles=None, vertex_normals=None):
    v = np.asarray(vertices, np.float32)
    n = None if vertex_normals is None else np.asarray(vertex_normals, np.float32)
    t = np.zeros((0, 3), np.int32) if triangles is None else np.asarray(triangles, np.int32)
    with open(path, "wb") as f:
        hdr = ["ply", "format binary_little_endian 1.0", f"element vertex {len(v)}",
               "property float x", "property float y", "property float z"]
        if n is not None:
            hdr += ["property float nx", "property float ny", "property float nz"]
        hdr += [f"element face {len(t)}", "property list uchar int vertex_indices", "end_header"]
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        f.write((np.concatenate([v, n], 1) if n is not None else v).astype("<f4").tobytes())
        if len(t):
            rec = np.empty(len(t), dtype=[("k", "u1"), ("i", "<i4", 3)])
            rec["k"], rec["i"] = 3, t
            f.write(rec.tobytes())
