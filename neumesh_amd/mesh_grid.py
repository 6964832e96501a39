"""MeshGrid -- host-side mirror of the reference's ``models/mesh_grid.py`` on the HIP library.

Same class / method names, argument meaning and return shapes as the reference
(models/mesh_grid.py:9-150) so callers (the NeuMesh field, the editing tools) do not change:

    MeshGrid(mesh, device, "frnn").compute_distance(xyz, indicator_vector, indicator_weight, K=8)
        -> (distance [N,1] f32, indices [N,K] int64, weights [N,K] f32)

The FRNN CUDA package the reference depends on is replaced by ``nm_grid_create`` / ``nm_knn`` /
``nm_compute_distance`` (include/neumesh_hip.h).  Under ``torch.no_grad()`` the whole method is one
fused kernel; with autograd enabled the K-NN (non-differentiable in the reference too,
mesh_grid.py:121-122) comes from the HIP kernel and the remaining arithmetic is expressed in
torch ops on the device so gradients w.r.t. xyz / indicator vectors flow as in the reference.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np
import torch

from . import _lib


def _as_device(device) -> torch.device:
    if isinstance(device, int):
        return torch.device("cuda", device)
    return torch.device(device)


class GridHandle:
    """Owns an nm_grid_t."""

    def __init__(self, vertices: torch.Tensor, leaf_level: int = 0):
        import os
        lib = _lib.load()
        leaf_level = int(os.environ.get("NEUMESH_LEAF_LEVEL", leaf_level))  # tuning knob; 0 = automatic
        if not vertices.is_cuda:
            raise _lib.NeuMeshHipError("MeshGrid needs a CUDA/HIP device tensor (no CPU fallback)")
        v = vertices.detach().to(torch.float32).contiguous()
        h = C.c_void_p()
        with torch.cuda.device(v.device):
            _lib.check(lib.nm_grid_create(_lib.ptr(v), v.shape[0], leaf_level, _lib.current_stream(v.device), C.byref(h)),
                       "nm_grid_create")
        self._h = h
        self.device = v.device
        self.num_vertices = int(v.shape[0])
        self._budget = None   # NEUMESH_KNN_BUDGET as last handed to the library (the library itself reads no environment)
        self._budget_lock = threading.Lock()

    @property
    def handle(self):
        b = os.environ.get("NEUMESH_KNN_BUDGET")   # tuning / test knob of the small-launch hand-over (nm_grid_set_option)
        if b != self._budget:                       # (changed since it was last handed over: rare; one string compare per access otherwise)
            with self._budget_lock:                 # nn.DataParallel threads may share a handle
                if b != self._budget:
                    try:
                        value = int(b) if b not in (None, "") else -1
                    except ValueError:              # a malformed value is "not set", as for every other NEUMESH_* knob
                        value = -1
                    _lib.check(_lib.load().nm_grid_set_option(self._h, _lib.GRID_DEFER_BUDGET, value), "nm_grid_set_option")
                    self._budget = b
        return self._h

    def trim(self):
        """Free the index's deferral scratch (33.7 MB per stream that ran small point-wise launches on it).  Call with the device idle."""
        _lib.check(_lib.load().nm_grid_set_option(self._h, _lib.GRID_TRIM, 0), "nm_grid_set_option")

    def info(self) -> dict:
        gi = _lib.GridInfo()
        _lib.check(_lib.load().nm_grid_get_info(self._h, C.byref(gi)), "nm_grid_get_info")
        return {"num_vertices": gi.num_vertices, "leaf_level": gi.leaf_level, "occupied_leaves": gi.occupied_leaves,
                "origin": tuple(gi.origin), "root_size": gi.root_size, "device_bytes": gi.device_bytes, "num_nodes": gi.num_nodes}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load(require_device=False).nm_grid_destroy(self._h)
                self._h = None
        except Exception:
            pass


def knn(grid: GridHandle, xyz: torch.Tensor, K: int):
    """Exact K nearest vertices: (idx int64 [Q,K], d2 f32 [Q,K]), ascending (d2, index)."""
    lib = _lib.load()
    q = xyz.detach().to(torch.float32).reshape(-1, 3).contiguous()
    Q = q.shape[0]
    idx = torch.empty((Q, K), dtype=torch.int64, device=q.device)
    d2 = torch.empty((Q, K), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _lib.check(lib.nm_knn(grid.handle, _lib.ptr(q), Q, K, _lib.ptr(idx), _lib.ptr(d2), _lib.current_stream(q.device)), "nm_knn")
    return idx, d2


class MeshPrimitive:
    """models/mesh_grid.py:9-43.  Ray casting against the mesh needs open3d's RaycastingScene and
    is only used by the interactive editing tools (out of the render hot path)."""

    def __init__(self, mesh):
        self.mesh = mesh
        if hasattr(mesh, "compute_vertex_normals"):
            self.mesh.compute_vertex_normals()
        self.scene = None

    def cast_ray(self, rays_o, rays_d):
        raise NotImplementedError("cast_ray needs open3d.t.geometry.RaycastingScene (editing tools only)")

    def get_number_of_vertices(self):
        return len(self.mesh.vertices)


class MeshGrid(MeshPrimitive):
    def __init__(self, mesh, device, distance_method="frnn"):
        super().__init__(mesh)
        dev = _as_device(device)
        self.vertices = torch.from_numpy(np.asarray(mesh.vertices, dtype=np.float32).copy()).to(dev)
        self.vertex_normals = torch.from_numpy(np.asarray(mesh.vertex_normals, dtype=np.float32).copy()).to(dev)
        self.grid = GridHandle(self.vertices)  # the reference caches FRNN's grid here (mesh_grid.py:64-74)
        self.distance_method = distance_method
        self.device = dev
        self._siblings = {}   # per-device copies for nn.DataParallel replicas (on_device)

    def on_device(self, device):
        """This mesh index on another device (own vertex copy + own octree), built once and cached: an nn.DataParallel
        replica of a NeuMesh on cuda:k must not launch kernels against the index that lives on cuda:0."""
        dev = _as_device(device)
        if dev == self.device:
            return self
        g = self._siblings.get(dev)
        if g is None:
            g = MeshGrid.__new__(MeshGrid)
            MeshPrimitive.__init__(g, self.mesh)
            g.vertices, g.vertex_normals = self.vertices.to(dev), self.vertex_normals.to(dev)
            g.grid = GridHandle(g.vertices)
            g.distance_method, g.device, g._siblings = self.distance_method, dev, {}
            self._siblings[dev] = g
        return g

    def compute_distance(self, xyz, indicator_vector=None, indicator_weight=0.1, K=8):
        if self.distance_method == "frnn":
            return self.compute_distance_frnn(xyz, K, indicator_vector=indicator_vector, indicator_weight=indicator_weight)
        raise NotImplementedError

    def compute_distance_frnn(self, xyz, K=8, indicator_vector=None, indicator_weight=0.1, want_grad=False):
        """(N,3) -> distance (N,1), indices (N,K) int64, weights (N,K); see module docstring.
        want_grad (no_grad path only) additionally returns d distance / d xyz (N,3)."""
        indicator = self.vertex_normals if indicator_vector is None else indicator_vector
        needs_graph = torch.is_grad_enabled() and (
            xyz.requires_grad or indicator.requires_grad
            or (torch.is_tensor(indicator_weight) and indicator_weight.requires_grad))
        if needs_graph:
            if want_grad:
                raise ValueError("want_grad is only meaningful without autograd")
            return self._compute_distance_autograd(xyz, K, indicator, indicator_weight)
        if K != 8:  # the fused kernel is built for the reference's K = 8 (mesh_grid.py:77); any other K <= 32: nm_knn + torch ops
            if want_grad:
                raise ValueError("want_grad needs K = 8 (fused kernel)")
            with torch.no_grad():
                return self._compute_distance_autograd(xyz, K, indicator, indicator_weight)
        lib = _lib.load()
        q = xyz.detach().to(torch.float32).reshape(-1, 3).contiguous()
        Q = q.shape[0]
        ind = indicator.detach().to(torch.float32).contiguous()
        w1 = float(indicator_weight)
        ds = torch.empty((Q, 1), dtype=torch.float32, device=q.device)
        idx = torch.empty((Q, K), dtype=torch.int64, device=q.device)
        w = torch.empty((Q, K), dtype=torch.float32, device=q.device)
        g = torch.empty((Q, 3), dtype=torch.float32, device=q.device) if want_grad else None
        with torch.cuda.device(q.device):
            _lib.check(lib.nm_compute_distance(self.grid.handle, _lib.ptr(q), Q, _lib.ptr(ind), w1, K, _lib.ptr(ds),
                                               _lib.ptr(idx), _lib.ptr(w), _lib.ptr(g), _lib.current_stream(q.device)),
                       "nm_compute_distance")
        return (ds, idx, w, g) if want_grad else (ds, idx, w)

    def compute_distance_interpolate(self, xyz, features, indicator_vector=None, indicator_weight=0.1):
        """compute_distance_frnn(xyz) followed by interpolation(features, indices, weights)
        (mesh_grid.py:88-144 + neumesh.py:11-13) as ONE kernel (nm_distance_interpolate): the wave that
        found the neighbours gathers their rows.  features: (V, dim), dim % 4 == 0 (any width).
        Inference only.  Returns (distance (N,1), indices (N,8), weights (N,8), interpolated (N,dim))."""
        if torch.is_grad_enabled() and (xyz.requires_grad or features.requires_grad):
            raise NotImplementedError("compute_distance_interpolate is the fused inference kernel; with autograd use "
                                      "compute_distance + interpolation")
        lib = _lib.load()
        indicator = self.vertex_normals if indicator_vector is None else indicator_vector
        q = xyz.detach().to(torch.float32).reshape(-1, 3).contiguous()
        tab = features.detach().to(torch.float32).contiguous()
        if tab.dim() != 2 or tab.shape[0] != self.vertices.shape[0]:
            raise ValueError("features must be (V, dim)")
        Q, dim = q.shape[0], tab.shape[1]
        ind = indicator.detach().to(torch.float32).contiguous()
        ds = torch.empty((Q, 1), dtype=torch.float32, device=q.device)
        idx = torch.empty((Q, 8), dtype=torch.int64, device=q.device)
        w = torch.empty((Q, 8), dtype=torch.float32, device=q.device)
        feat = torch.empty((Q, dim), dtype=torch.float32, device=q.device)
        with torch.cuda.device(q.device):
            _lib.check(lib.nm_distance_interpolate(self.grid.handle, _lib.ptr(q), Q, _lib.ptr(ind), float(indicator_weight),
                                                   _lib.ptr(tab), dim, _lib.ptr(ds), _lib.ptr(idx), _lib.ptr(w), _lib.ptr(feat),
                                                   _lib.current_stream(q.device)), "nm_distance_interpolate")
        return ds, idx, w, feat

    def _compute_distance_autograd(self, xyz, K, indicator, indicator_weight):
        # K-NN on the HIP kernel (detached, like mesh_grid.py:121-122), the rest differentiable
        idx, d2 = knn(self.grid, xyz, K)
        dis = d2.sqrt()
        weights = 1 / (dis + 1e-7)
        weights = weights / weights.sum(dim=-1, keepdim=True)
        diff = xyz.reshape(-1, 3).unsqueeze(-2) - self.vertices[idx]
        r = torch.norm(diff, dim=-1, keepdim=True)
        w1 = indicator_weight
        mid = (indicator[idx] * w1 + diff * r) / (w1 + r)
        distance = (weights.unsqueeze(-1) * (diff * mid).sum(dim=-1, keepdim=True)).sum(dim=-2)
        return distance, idx, weights

    def get_vertex_normal_torch(self):
        return self.vertex_normals

    def get_vertices_torch(self):
        return self.vertices
