"""Texture-editing wrapper -- host-side mirror of the reference's
``editing/texture_neumesh/texture_neumesh.py`` (class TextureEditableNeuMesh, :7-122): a main NeuMesh whose
painted vertices take their colour from reference model(s), blended by how much of a point's interpolation
weight sits on painted vertices.  Same constructor arguments, buffers and methods, so the reference's texture
renderers (editing/texture_neumesh/texture_renderer.py:73-86) and this package's ``volume_render`` (staged path,
renderer.render_rays_staged) drive it unchanged.

``forward`` is the hot call (every mid-point of every ray).  Under ``torch.no_grad()`` it is: ONE fused HIP call
for the main model (K-NN + projected distance + gathers + geometry MLP with nabla + colour MLP, returning
sdf / rgb / nabla / ds / indices / weights), a few element-wise device ops for the blend weights on the [P,8]
neighbour lists, and one fused colour call (gather of the edited codes + colour MLP) per reference model on the
painted points only.  With autograd enabled the same formulas run through the models' differentiable forms.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def transform_direction(rotation, dirs):
    """utils/geo_util.py:78-89: rotate (...,3) directions by a (3,3) matrix."""
    return torch.matmul(rotation, dirs.unsqueeze(-1)).squeeze(-1)


class TextureEditableNeuMesh(nn.Module):
    def __init__(self, main_model, ref_models, main_editing_masks, main_editing_colorfeats, T_r_m_list=None):
        super().__init__()
        self.main_model = main_model
        self.ref_models = nn.ModuleList(ref_models)
        self.register_buffer("main_editing_masks", main_editing_masks)            # [n_ref, V] bool
        self.register_buffer("main_editing_colorfeats", main_editing_colorfeats)  # [V, color_dim]
        if T_r_m_list is not None:   # reference-from-main rigid transforms: rotations for directions / nablas
            self.register_buffer("rot_s_m", torch.stack([T[:3, :3] for T in T_r_m_list], dim=0))
            self.register_buffer("t_s_m", torch.stack([T[:3, 3] for T in T_r_m_list], dim=0))
        else:
            self.rot_s_m = None
            self.t_s_m = None
        self.enable_nablas_input = main_model.enable_nablas_input

    # ---- what the renderer needs besides forward: delegated to the main model (texture_neumesh.py:41-51)
    def compute_distance(self, xyz):
        return self.main_model.compute_distance(xyz)

    def forward_s(self):
        return self.main_model.forward_s()

    def forward_density_only(self, xyz):
        return self.main_model.forward_density_only(xyz)

    def forward_with_nablas(self, xyz: torch.Tensor):
        return self.main_model.forward_with_nablas(xyz)

    def _main_query(self, xyz, view_dirs, need_nablas):
        """(sdf, nabla, ds, indices, weights, main colour): texture_neumesh.py:64-78."""
        m = self.main_model
        if not torch.is_grad_enabled() and need_nablas and hasattr(m, "_fused_forward") and m.fused_supported():
            sdf, rgb, nabla, ds, idx, w = m._fused_forward(xyz, view_dirs, True)   # one fused HIP call
            return sdf, nabla, ds, idx, w, rgb
        sdf, nabla, ds, idx, w = m.forward(xyz, view_dirs, need_nablas=need_nablas, nablas_only=True, return_ds=True)
        return sdf, nabla, ds, idx, w, m.forward_color(ds, view_dirs, m.color_features, indices=idx, weights=w, nabla=nabla)

    def forward(self, xyz: torch.Tensor, view_dirs: torch.Tensor, need_nablas=True, nablas_only=False):
        """xyz, view_dirs: (...,3) -> (sdf (...,1), blended colour (...,3)); texture_neumesh.py:53-122."""
        sdf, nabla, ds, idx, w, colors = self._main_query(xyz, view_dirs, need_nablas)
        blend = colors.clone()
        for i, ref_model in enumerate(self.ref_models):
            painted = self.main_editing_masks[i][idx]                    # (...,8): is neighbour k painted?
            on_paint = torch.sum(w * painted, dim=-1)                    # interpolation weight on painted vertices
            on_rest = torch.sum(w * (painted == False), dim=-1)          # noqa: E712  (as the reference writes it)
            region = on_paint > 0
            total = on_paint + on_rest
            a_paint, a_rest = (on_paint / total)[region], (on_rest / total)[region]
            ref_w = w * painted
            ref_w = ref_w / (torch.sum(ref_w, dim=-1, keepdim=True) + 1e-8)
            if self.rot_s_m is not None:
                ref_dir, ref_nabla = transform_direction(self.rot_s_m[i], view_dirs), transform_direction(self.rot_s_m[i], nabla)
            else:
                ref_dir, ref_nabla = view_dirs, nabla
            if bool(torch.any(region)):
                ref_color = ref_model.forward_color(ds[region], ref_dir.expand_as(xyz)[region], self.main_editing_colorfeats,
                                                    indices=idx[region], weights=ref_w[region], nabla=ref_nabla[region])
                blend[region] = blend[region] * a_rest.unsqueeze(-1) + ref_color * a_paint.unsqueeze(-1)
        return sdf, blend


# ---- geometry editing: the model follows a deformed mesh (editing/render_geometry_editing.py:19-67)

def cos_between_vectors(x, y, do_clamp=True):
    """render_geometry_editing.py:19-34: cosine of the angle between (...,3) vectors, clamped to [-1, 1]."""
    c = torch.sum(x * y, dim=-1) / (torch.linalg.norm(x, dim=-1) * torch.linalg.norm(y, dim=-1))
    return torch.clamp(c, -1, 1) if do_clamp else c


def angle_axis_to_rotation_matrix(angle_axis: torch.Tensor) -> torch.Tensor:
    """(N,3) rotation vectors -> (N,3,3).  The reference takes this from kornia 0.6.3 (environment.yml:42;
    kornia.geometry.conversions.angle_axis_to_rotation_matrix, itself after ceres/rotation.h): Rodrigues' formula with the
    axis normalised as v / (theta + 1e-6) where theta^2 = |v|^2 > 1e-6, and the first-order form I + [v]x below that.  Restated
    here on device ops (one call per mesh edit, N = V) because the two quirks are visible in the result: the 1e-6 in the
    normalisation, and that a zero vector gives exactly the identity."""
    v = angle_axis
    theta2 = (v * v).sum(-1, keepdim=True)
    theta = torch.sqrt(theta2)
    w = v / (theta + 1e-6)
    wx, wy, wz = w[:, 0:1], w[:, 1:2], w[:, 2:3]
    c, s = torch.cos(theta), torch.sin(theta)
    k = 1.0 - c
    full = torch.cat([c + wx * wx * k, wx * wy * k - wz * s, wy * s + wx * wz * k,
                      wz * s + wx * wy * k, c + wy * wy * k, -wx * s + wy * wz * k,
                      -wy * s + wx * wz * k, wx * s + wy * wz * k, c + wz * wz * k], dim=1).view(-1, 3, 3)
    rx, ry, rz = v[:, 0:1], v[:, 1:2], v[:, 2:3]
    one = torch.ones_like(rx)
    small = torch.cat([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    big = (theta2 > 1e-6).view(-1, 1, 1).to(v.dtype)
    return big * full + (1.0 - big) * small


def deform_model(deformed_mesh, model, device, fix_indicator=False):
    """render_geometry_editing.py:37-67, same arguments: the model's mesh index is rebuilt on the deformed mesh (octree built on the
    device, < 1 ms) and, unless `fix_indicator`, every indicator vector is turned by the rotation that takes the vertex's old normal
    to its new one.  As in the reference the rotation vector is `cross(n_old, n_new) * acos(cos)` with the cross product NOT
    normalised (its length is sin(angle), so the turn is by angle * sin(angle)), a normal that flips exactly (cos == -1) negates
    the indicator, and the result replaces `model.indicator_vector` as a new nn.Parameter."""
    from .mesh_grid import MeshGrid
    new_grid = MeshGrid(deformed_mesh, device, distance_method=model.mesh_grid.distance_method)
    if not fix_indicator:
        with torch.no_grad():
            n_old = model.mesh_grid.get_vertex_normal_torch()
            n_new = new_grid.get_vertex_normal_torch().to(n_old.device)
            axis = torch.cross(n_old, n_new, dim=-1)
            cos_theta = cos_between_vectors(n_old, n_new)
            flipped = cos_theta == -1
            rot = angle_axis_to_rotation_matrix(axis * torch.acos(cos_theta).unsqueeze(-1))
            ind = torch.matmul(rot, model.indicator_vector.detach().unsqueeze(-1)).squeeze(-1)
            ind[flipped] *= -1
        model.indicator_vector = nn.Parameter(ind)
    model.mesh_grid = new_grid
