"""oracle/editing.py -- TEST INFRASTRUCTURE ONLY (CPU parity oracle, numpy fp32).

Restates the two editing consumers of the field:

* TextureEditableNeuMesh.forward          editing/texture_neumesh/texture_neumesh.py:53-122
  (+ the delegations :41-51, the transforms :21-37, utils/geo_util.py:78-89)
* deform_model's indicator rotation       editing/render_geometry_editing.py:19-34, :44-65, with
  kornia.geometry.conversions.angle_axis_to_rotation_matrix (kornia 0.6.3, environment.yml:42 -- third-party, absent from
  /root/reference: its published algorithm, Rodrigues' formula after ceres/rotation.h, is restated below)

Pinned by tests/golden/texture_edit_v3000.npz and deform_v3000.npz, which oracle/gen_golden.py produced by running the
reference's own classes / functions (tests/test_oracle.py).
"""
from __future__ import annotations

import numpy as np

from .field import F32, OracleField


class OracleTextureEdit:
    """numpy twin of TextureEditableNeuMesh over OracleFields.  masks: [n_ref, V] bool; feats: [V, color_dim];
    T_list: None or n_ref 4x4 matrices (only their rotations are used, as in the reference)."""

    def __init__(self, main: OracleField, refs, masks, feats, T_list=None):
        self.main, self.refs = main, list(refs)
        self.masks = np.asarray(masks, bool)
        self.feats = np.ascontiguousarray(feats, F32)
        self.rot = None if T_list is None else [np.ascontiguousarray(T, F32)[:3, :3] for T in T_list]

    # texture_neumesh.py:41-51
    def compute_distance(self, xyz):
        return self.main.compute_distance(xyz)

    def forward_s(self):
        return self.main.forward_s()

    def forward_density_only(self, xyz):
        return self.main.forward_density_only(xyz)

    def forward_with_nablas(self, xyz):
        return self.main.forward_with_nablas(xyz)

    def forward(self, xyz, view_dirs):
        """:53-122 -> (sdf, blended colour, nabla)."""
        shp = xyz.shape[:-1]
        sdf, colors, nabla, ds, idx, w = self.main.forward(xyz.reshape(-1, 3), view_dirs.reshape(-1, 3), return_ds=True)   # :64-78
        view = np.ascontiguousarray(view_dirs, F32).reshape(-1, 3)
        blend = colors.copy()
        for i, ref in enumerate(self.refs):
            painted = self.masks[i][idx]                                           # :85
            pw = np.sum(w * painted, axis=-1, dtype=F32)
            uw = np.sum(w * (painted == False), axis=-1, dtype=F32)                # noqa: E712  (:86-88)
            region = pw > 0
            tot = pw + uw
            with np.errstate(invalid="ignore", divide="ignore"):
                pw, uw = (pw / tot)[region], (uw / tot)[region]                    # :90-94
            rw = (w * painted).astype(F32)
            rw = rw / (np.sum(rw, axis=-1, keepdims=True, dtype=F32) + F32(1e-8))  # :96-97
            if self.rot is not None:                                               # :100-106 (geo_util.transform_direction)
                rdir, rnab = (view @ self.rot[i].T).astype(F32), (nabla @ self.rot[i].T).astype(F32)
            else:
                rdir, rnab = view, nabla
            if region.any():                                                       # :107-120
                rc = ref.forward_color(ds[region], rdir[region], self.feats, idx[region], rw[region], rnab[region])
                blend[region] = blend[region] * uw[:, None] + rc * pw[:, None]
        return sdf.reshape(*shp, 1), blend.reshape(*shp, 3), nabla.reshape(*shp, 3)


def angle_axis_to_rotation_matrix(v: np.ndarray) -> np.ndarray:
    """kornia 0.6.3: theta^2 = |v|^2; above 1e-6: axis v / (theta + 1e-6), R = c I + (1-c) w w^T + s [w]x; else I + [v]x."""
    v = np.ascontiguousarray(v, F32)
    th2 = np.sum(v * v, axis=-1, dtype=F32)
    th = np.sqrt(th2)
    w = v / (th + F32(1e-6))[:, None]
    x, y, z = w[:, 0], w[:, 1], w[:, 2]
    c, s = np.cos(th), np.sin(th)
    k = F32(1.0) - c
    big = np.stack([c + x * x * k, x * y * k - z * s, y * s + x * z * k,
                    z * s + x * y * k, c + y * y * k, -x * s + y * z * k,
                    -y * s + x * z * k, x * s + y * z * k, c + z * z * k], -1).reshape(-1, 3, 3)
    one = np.ones_like(th)
    small = np.stack([one, -v[:, 2], v[:, 1], v[:, 2], one, -v[:, 0], -v[:, 1], v[:, 0], one], -1).reshape(-1, 3, 3)
    return np.where((th2 > F32(1e-6))[:, None, None], big, small).astype(F32)


def deform_indicator(n_old: np.ndarray, n_new: np.ndarray, indicator: np.ndarray) -> np.ndarray:
    """render_geometry_editing.py:44-65: rotation vector cross(n_old, n_new) * acos(clamp(cos)) -- the cross product is not
    normalised --, indicator turned by it, negated where cos == -1 exactly."""
    n_old, n_new, indicator = (np.ascontiguousarray(a, F32) for a in (n_old, n_new, indicator))
    axis = np.cross(n_old, n_new).astype(F32)
    cos = np.sum(n_old * n_new, axis=-1, dtype=F32) / (np.linalg.norm(n_old, axis=-1) * np.linalg.norm(n_new, axis=-1)).astype(F32)
    cos = np.clip(cos, F32(-1), F32(1))
    R = angle_axis_to_rotation_matrix(axis * np.arccos(cos)[:, None])
    out = np.einsum("vij,vj->vi", R, indicator).astype(F32)
    out[cos == -1] *= -1
    return out
