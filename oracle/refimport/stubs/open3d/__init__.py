"""Minimal `open3d` for the reference's mesh handling (models/mesh_grid.py:19-24,60-63,
models/frameworks/neumesh/__init__.py:14).  `io.read_triangle_mesh(key)` returns a mesh that the
harness registered under `key` (there are no .ply files in the container)."""
import numpy as np

_REGISTRY = {}


class TriangleMesh:
    def __init__(self, vertices, vertex_normals, triangles=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.vertex_normals = np.asarray(vertex_normals, dtype=np.float64)
        self.triangles = np.zeros((0, 3), np.int32) if triangles is None else np.asarray(triangles)

    def compute_vertex_normals(self):
        return self  # normals are supplied analytically by the synthetic scene


def register_mesh(key, vertices, vertex_normals, triangles=None):
    _REGISTRY[key] = TriangleMesh(vertices, vertex_normals, triangles)


class _IO:
    @staticmethod
    def read_triangle_mesh(path):
        return _REGISTRY[path]


class _TTriangleMesh:
    @staticmethod
    def from_legacy(mesh):
        return mesh


class _RaycastingScene:
    def add_triangles(self, mesh):
        return 0

    def cast_rays(self, rays):
        raise NotImplementedError("open3d stub: no ray casting")


class _TGeometry:
    TriangleMesh = _TTriangleMesh
    RaycastingScene = _RaycastingScene


class _T:
    geometry = _TGeometry


io = _IO
t = _T
