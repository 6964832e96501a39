"""oracle/field.py -- TEST INFRASTRUCTURE ONLY (CPU parity oracle, numpy fp32).

Restates, function by function, the NeuMesh field the renderer queries:

* projected signed distance + K-NN weights   models/mesh_grid.py:88-144
* kNN-weighted code interpolation            models/frameworks/neumesh/neumesh.py:11-13
* positional encoding                        models/base.py:52-70, :73-87
* geometry MLP (weight-norm, softplus b=100) models/frameworks/neumesh/neumesh.py:204-237 (:76-86,:101)
* nabla = d sdf / d xyz                       neumesh.py:225-232 (autograd in the reference;
  here the closed form of SURVEY.md section 3.4 -- valid because indices/weights are detached at
  mesh_grid.py:121-122 -- checked against the reference's autograd by gen_golden.py)
* colour MLP                                 neumesh.py:239-260 (:93-102)
* forward / forward_density_only / forward_with_nablas / forward_s / compute_distance
                                             neumesh.py:113-171, :262-273

All arrays are float32; python scalars stay "weak" under numpy>=2 so no silent promotion.
"""
from __future__ import annotations

from dataclasses import dataclass, field as _dc_field
from typing import Callable, Dict, Optional

import numpy as np

from . import knn as _knn

F32 = np.float32


# --------------------------------------------------------------------------- helpers
def embed(x: np.ndarray, n_freqs: int) -> np.ndarray:
    """models/base.py:52-70 with get_embedder's settings (:73-87): include_input,
    log-sampled bands 2**linspace(0, L-1, L), per band [sin(x*f), cos(x*f)] over all dims."""
    if n_freqs < 0:
        return x
    out = [x]
    for j in range(n_freqs):
        f = F32(2.0 ** j)
        out.append(np.sin(x * f))
        out.append(np.cos(x * f))
    return np.concatenate(out, axis=-1).astype(F32)


def embed_out_dim(in_dim: int, n_freqs: int) -> int:
    return in_dim if n_freqs < 0 else in_dim * (1 + 2 * n_freqs)


def softplus100(x: np.ndarray) -> np.ndarray:
    """torch.nn.Softplus(beta=100) (threshold=20): x if x*beta > 20 else log1p(exp(x*beta))/beta."""
    xb = x * F32(100.0)
    with np.errstate(over="ignore"):
        soft = np.log1p(np.exp(xb)) / F32(100.0)
    return np.where(xb > F32(20.0), x, soft).astype(F32)


def softplus100_grad(x: np.ndarray) -> np.ndarray:
    """autograd of Softplus(beta=100): 1 above the threshold else z/(z+1), z=exp(x*beta)."""
    xb = x * F32(100.0)
    with np.errstate(over="ignore", invalid="ignore"):
        z = np.exp(xb)
        g = z / (z + F32(1.0))
    return np.where(xb > F32(20.0), F32(1.0), g).astype(F32)


def sigmoid(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """torch.nn.utils.weight_norm (dim=0): W = v * (g / ||v||_row)."""
    n = np.sqrt(np.sum(v.astype(F32) ** 2, axis=1, keepdims=True, dtype=F32))
    return (v * (g / n)).astype(F32)


# --------------------------------------------------------------------------- parameters
@dataclass
class FieldConfig:
    """Mirror of get_model()'s model_config (models/frameworks/neumesh/__init__.py:23-51)."""
    D_density: int = 3
    D_color: int = 4
    W: int = 256
    geometry_dim: int = 32
    color_dim: int = 32
    multires_view: int = 4
    multires_d: int = 8
    multires_fg: int = 2
    multires_ft: int = 2
    enable_nablas_input: bool = True
    speed_factor: float = 10.0
    learn_indicator_weight: bool = False
    K: int = 8


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.ascontiguousarray(x, dtype=F32)


@dataclass
class OracleField:
    """numpy twin of models/frameworks/neumesh/neumesh.py:NeuMesh + models/mesh_grid.py:MeshGrid."""
    vertices: np.ndarray
    state: Dict[str, np.ndarray]
    cfg: FieldConfig = _dc_field(default_factory=FieldConfig)
    knn_fn: Optional[Callable] = None  # (q, verts, K) -> (idx, d2); default brute force

    def __post_init__(self):
        self.vertices = _np(self.vertices)
        self.state = {k: _np(v) for k, v in self.state.items()}
        s = self.state
        c = self.cfg
        # geometry MLP: pts_linears.{0, 2.0, 3.0, ...} are weight-normed (neumesh.py:76-86)
        self.geo_W, self.geo_b = [], []
        for li in range(c.D_density):
            key = "pts_linears.0" if li == 0 else f"pts_linears.{li + 1}.0"
            self.geo_W.append(fold_weight_norm(s[key + ".weight_g"], s[key + ".weight_v"]))
            self.geo_b.append(s[key + ".bias"])
        self.den_W = fold_weight_norm(s["density_linear.weight_g"], s["density_linear.weight_v"])
        self.den_b = s["density_linear.bias"]
        # colour MLP: views_linears.{0, 2.0, 3.0, 4.0} plain Linear + ReLU (neumesh.py:93-100)
        self.col_W, self.col_b = [], []
        for li in range(c.D_color):
            key = "views_linears.0" if li == 0 else f"views_linears.{li + 1}.0"
            self.col_W.append(s[key + ".weight"])
            self.col_b.append(s[key + ".bias"])
        self.out_W = s["color_linear.0.weight"]
        self.out_b = s["color_linear.0.bias"]

    # ---- scalars
    def forward_s(self) -> np.float32:
        """neumesh.py:170-171."""
        return np.exp(self.state["ln_s"] * F32(self.cfg.speed_factor)).astype(F32)[0]

    def indicator_weight(self) -> np.float32:
        """neumesh.py:173-174 / :266-268 (0.1 unless learn_indicator_weight)."""
        if self.cfg.learn_indicator_weight:
            return sigmoid(self.state["indicator_weight_raw"])[0]
        return F32(0.1)

    # ---- mesh_grid.py:88-144
    def knn(self, xyz: np.ndarray):
        fn = self.knn_fn or _knn.knn_bruteforce
        return fn(xyz.reshape(-1, 3), self.vertices, self.cfg.K)

    def compute_distance(self, xyz: np.ndarray, want_grad: bool = False):
        """Returns ds [...,1], idx [...,K] int64, w [...,K] (and d ds/d xyz [...,3])."""
        shp = xyz.shape[:-1]
        x = np.ascontiguousarray(xyz, dtype=F32).reshape(-1, 3)
        idx, d2 = self.knn(x)
        dis = np.sqrt(d2)                                   # mesh_grid.py:123
        w = F32(1.0) / (dis + F32(1e-7))                    # :124
        w = w / np.sum(w, axis=-1, keepdims=True, dtype=F32)  # :125
        w1 = self.indicator_weight()
        ind = self.state["indicator_vector"]
        dir_vec = x[:, None, :] - self.vertices[idx]        # :134
        w2 = np.sqrt(np.sum(dir_vec * dir_vec, axis=-1, keepdims=True, dtype=F32))  # :135
        mid = (ind[idx] * w1 + dir_vec * w2) / (w1 + w2)    # :136
        f = np.sum(dir_vec * mid, axis=-1, keepdims=True, dtype=F32)  # :137-141
        ds = np.sum(w[..., None] * f, axis=-2, dtype=F32)   # :142 -> [Q,1]
        out = (ds.reshape(*shp, 1), idx.reshape(*shp, -1), w.reshape(*shp, -1))
        if not want_grad:
            return out
        # closed-form d ds / d xyz (weights & indices detached, mesh_grid.py:121-122)
        r = w2
        with np.errstate(divide="ignore", invalid="ignore"):
            u = np.where(r > 0, dir_vec / r, F32(0.0)).astype(F32)
        a = np.sum(dir_vec * ind[idx], axis=-1, keepdims=True, dtype=F32)
        num = (ind[idx] * w1 + F32(3.0) * r * r * u) * (w1 + r) - (w1 * a + r * r * r) * u
        dfdx = num / ((w1 + r) * (w1 + r))
        g = np.sum(w[..., None] * dfdx, axis=-2, dtype=F32)
        return out + (g.reshape(*shp, 3),)

    # ---- neumesh.py:11-13
    @staticmethod
    def interpolation(features, idx, w):
        return np.sum(features[idx] * w[..., None], axis=-2, dtype=F32)

    # ---- neumesh.py:204-237
    def _geometry(self, ds, idx, w, want_tangent: bool):
        c = self.cfg
        d_emb = embed(ds, c.multires_d)
        fg = self.interpolation(self.state["geometry_features"], idx, w)
        h = np.concatenate([d_emb, embed(fg, c.multires_fg)], axis=-1)
        t = None
        if want_tangent:
            # d(d_emb)/d(ds): [1, f cos(f ds), -f sin(f ds), ...] then zeros for fg_emb
            td = [np.ones_like(ds)]
            for j in range(c.multires_d):
                fj = F32(2.0 ** j)
                td.append(fj * np.cos(ds * fj))
                td.append(-fj * np.sin(ds * fj))
            t = np.concatenate(td + [np.zeros(h.shape[:-1] + (h.shape[-1] - len(td),), F32)], axis=-1)
        for W, b in zip(self.geo_W, self.geo_b):
            z = (h @ W.T + b).astype(F32)
            h = softplus100(z)
            if want_tangent:
                t = ((t @ W.T) * softplus100_grad(z)).astype(F32)
        sdf = (h @ self.den_W.T + self.den_b).astype(F32)
        dsdf_dds = (t @ self.den_W.T).astype(F32) if want_tangent else None
        return sdf, dsdf_dds, d_emb

    def forward_density_only(self, xyz):
        """neumesh.py:140-145."""
        ds, idx, w = self.compute_distance(xyz)
        sdf, _, _ = self._geometry(ds, idx, w, False)
        return sdf

    def forward_with_nablas(self, xyz):
        """neumesh.py:147-154; nabla = (d sdf/d ds) * (d ds/d xyz)."""
        ds, idx, w, g = self.compute_distance(xyz, want_grad=True)
        sdf, dsdf, _ = self._geometry(ds, idx, w, True)
        return sdf, (dsdf * g).astype(F32)

    # ---- neumesh.py:239-260
    def _color(self, d_emb, view_dirs, color_features, idx, w, nabla):
        c = self.cfg
        parts = []
        if c.enable_nablas_input:
            parts.append(nabla)
        parts.append(d_emb)
        parts.append(embed(np.ascontiguousarray(view_dirs, dtype=F32), c.multires_view))
        parts.append(embed(self.interpolation(color_features, idx, w), c.multires_ft))
        h = np.concatenate(parts, axis=-1).astype(F32)
        for W, b in zip(self.col_W, self.col_b):
            h = np.maximum(h @ W.T + b, F32(0.0)).astype(F32)
        return sigmoid((h @ self.out_W.T + self.out_b).astype(F32))

    def forward_color(self, d, view_dirs, color_features, idx, w, nabla):
        """neumesh.py:156-168."""
        return self._color(embed(d, self.cfg.multires_d), view_dirs, _np(color_features), idx, w, nabla)

    def forward(self, xyz, view_dirs, return_ds: bool = False):
        """neumesh.py:113-138 with need_nablas=True (the renderer's call, renderer.py:279-282)."""
        ds, idx, w, g = self.compute_distance(xyz, want_grad=True)
        sdf, dsdf, d_emb = self._geometry(ds, idx, w, True)
        nabla = (dsdf * g).astype(F32)
        rgb = self._color(d_emb, view_dirs, self.state["color_features"], idx, w, nabla)
        if return_ds:
            return sdf, rgb, nabla, ds, idx, w
        return sdf, rgb, nabla
