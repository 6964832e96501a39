"""Shared helpers of the test-suite: scene / model construction from the golden fixtures."""
from __future__ import annotations

import os

import numpy as np

from neumesh_amd import synthetic

DEFAULT_PRECISION = os.environ.get("NEUMESH_MLP_PRECISION", "f16x2s")   # the library default (neumesh_amd/neumesh.py)
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

MODEL_CFG = dict(D_density=3, D_color=4, W=256, geometry_dim=32, color_dim=32, multires_view=4, multires_d=8,
                 multires_fg=2, multires_ft=2, enable_nablas_input=True, speed_factor=10.0,
                 learn_indicator_weight=False)


def golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def scene_mesh(V: int, dup: int = 0) -> synthetic.SyntheticMesh:
    """The mesh the fixtures were generated on (oracle/gen_golden.py)."""
    mesh = synthetic.fibonacci_blob(V)
    if dup:
        mesh = synthetic.SyntheticMesh(np.concatenate([mesh.vertices, mesh.vertices[:dup]]),
                                       np.concatenate([mesh.vertex_normals, mesh.vertex_normals[:dup]]))
    return mesh


def scene_state(mesh, mlp_seed_file: str = "model_seed0") -> dict:
    """Full NeuMesh state dict (numpy) of the fixture scenes: MLP weights stored in the fixture
    (reference constructor under torch.manual_seed(0)), codes / indicator vectors re-seeded."""
    V = mesh.num_vertices
    sd = {k: v for k, v in golden(mlp_seed_file).items()}
    sd["geometry_features"] = synthetic.random_codes(V, MODEL_CFG["geometry_dim"], 1)
    sd["color_features"] = synthetic.random_codes(V, MODEL_CFG["color_dim"], 2)
    sd["indicator_vector"] = synthetic.noisy_indicator(mesh.vertex_normals, 3)
    return sd


def surface_state(mesh) -> dict:
    """The scene WITH a surface (tests/golden/render_v140k_surf.npz): the fixture weights re-shaped by
    synthetic.surface_mlp_state (unit 0 of the geometry layers carries ds, sdf = ds + a code-driven bump), s = 400."""
    sd = scene_state(mesh)
    sd.update(synthetic.surface_mlp_state({k: v for k, v in golden("model_seed0").items()}))
    return sd


def state_digest(state) -> str:
    """sha256 over the MLP tensors (sorted keys, fp32 bytes) -- as oracle/gen_golden.py:state_digest."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k], dtype=np.float32).tobytes())
    return h.hexdigest()


class MeshObj:
    """Duck-type of the open3d mesh the reference passes to MeshGrid."""

    def __init__(self, mesh):
        self.vertices = np.asarray(mesh.vertices, np.float64)
        self.vertex_normals = np.asarray(mesh.vertex_normals, np.float64)

    def compute_vertex_normals(self):
        return self


def make_oracle(mesh, state):
    from oracle import field as ofield
    cfg = ofield.FieldConfig(speed_factor=MODEL_CFG["speed_factor"], learn_indicator_weight=False, enable_nablas_input=True)
    return ofield.OracleField(mesh.vertices, state, cfg)


def make_model(mesh, state, device):
    """neumesh_amd.NeuMesh on `device`, loaded through load_state_dict (checks ckpt-layout compat)."""
    import torch
    from neumesh_amd import MeshGrid, NeuMesh
    grid = MeshGrid(MeshObj(mesh), device)
    model = NeuMesh(grid, **MODEL_CFG)
    missing = model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model.to(device).eval()


class StubTeacher:
    """The analytic stand-in teacher the train-step fixture was generated with (oracle/gen_golden.py)."""

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def __call__(self, xyz, dirs):
        import torch
        return torch.linalg.norm(xyz, dim=-1) - 0.75, torch.sigmoid(2.0 * dirs + xyz)


def edit_oracle(mesh, state, n_ref: int, rotated: bool):
    """oracle.editing.OracleTextureEdit of the texture_edit_v3000 scene (synthetic.edit_scene / reference_color_state)."""
    from oracle import editing as oedit
    masks, feats, T_list = synthetic.edit_scene(mesh.vertices, n_ref, rotated)
    mlp = {k: v for k, v in golden("model_seed0").items()}
    refs = [make_oracle(mesh, {**state, **synthetic.reference_color_state(mlp, i)}) for i in range(n_ref)]
    return oedit.OracleTextureEdit(make_oracle(mesh, state), refs, masks, feats, T_list)


def edit_model(mesh, state, n_ref: int, rotated: bool, device):
    """neumesh_amd.editing.TextureEditableNeuMesh of the same scene on `device` (+ the main model)."""
    import torch
    from neumesh_amd.editing import TextureEditableNeuMesh
    masks, feats, T_list = synthetic.edit_scene(mesh.vertices, n_ref, rotated)
    mlp = {k: v for k, v in golden("model_seed0").items()}
    main = make_model(mesh, state, device)
    refs = [make_model(mesh, {**state, **synthetic.reference_color_state(mlp, i)}, device) for i in range(n_ref)]
    T = None if T_list is None else [torch.from_numpy(t).to(device) for t in T_list]
    wrap = TextureEditableNeuMesh(main, refs, torch.from_numpy(masks).to(device), torch.from_numpy(feats).to(device), T)
    return wrap.eval(), main


def trained_state(name: str = "trained_v140k") -> dict:
    """Every tensor of the TRAINED checkpoint tests/golden/<name>.pt (tools/train_field.py; utils/checkpoints.py layout, read the way
    render.py:287-288 does: the "model" entry) as numpy."""
    import torch
    sd = torch.load(os.path.join(GOLDEN, name + ".pt"), map_location="cpu")["model"]
    return {k: v.numpy() for k, v in sd.items()}


def field_margins(model, rays_o, rays_d, depths, max_points: int = 1 << 17) -> dict:
    """What the split-half f16 kernels' operands look like on THIS weight set (VERDICT r5 item 1 iv), evaluated with torch ops in fp32 on the
    sample points rays_o + normalize(rays_d) * depths (neighbours / weights from the product's compute_distance):
      * value rows: the largest |operand| any k-loop reads -- embedded inputs, activations in the kernels' log2 units (S y, S = 100 / ln 2),
        packed weights (layer 0 x S) -- against the fp16 range 65504;
      * tangent rows (one-accumulator mode): the largest |2^-8 S d y_l / d ds| and 2^-8 x the embedding derivatives, against 65504 above and
        the residual halves' absolute resolution 2^-25 below (NM_H2_TANGENT_SCALE_1ACC in nm_mlp_h2.h was chosen on untrained weights);
      * the share of 4-dim code chunks whose sin / cos arguments leave the fast polynomial range (NM_SINCOS_FAST_MAX = 1e5, nm_mlp.h);
      * s = exp(ln_s x speed_factor)."""
    import torch
    import torch.nn.functional as F
    S, TS = 144.26950408889634, 2.0 ** -8
    with torch.no_grad():
        dn = F.normalize(rays_d, dim=-1)
        xyz = (rays_o[:, None, :] + dn[:, None, :] * depths[..., None]).reshape(-1, 3)
        if xyz.shape[0] > max_points:
            xyz = xyz[:: -(-xyz.shape[0] // max_points)]
        ds, idx, w = model.compute_distance(xyz)
        fg = (model.geometry_features[idx] * w.unsqueeze(-1)).sum(-2)
        ft = (model.color_features[idx] * w.unsqueeze(-1)).sum(-2)

        def folded(m):   # weight_norm, dim 0
            v, g = m.weight_v, m.weight_g
            return g * v / v.norm(dim=1, keepdim=True)
        d_emb, fg_emb = model.embed_fn_d(ds), model.embed_fn_fg(fg)
        L = model.embed_fn_d.n_freqs
        # d(embedding of ds) / d ds: [1, f cos(f ds), -f sin(f ds), ...]
        t_parts = [torch.ones_like(ds)]
        for j in range(L):
            f = float(2 ** j)
            t_parts += [f * torch.cos(ds * f), -f * torch.sin(ds * f)]
        x = torch.cat([d_emb, fg_emb], -1)
        t = torch.cat(t_parts + [torch.zeros_like(fg_emb)], -1)
        out = {"s": float(model.forward_s()), "points": int(xyz.shape[0]),
               "max_abs_code_interpolated": float(torch.maximum(fg.abs().max(), ft.abs().max())),
               "max_abs_code_table": float(torch.maximum(model.geometry_features.abs().max(), model.color_features.abs().max()))}
        op_v, op_t, t_min_scale = float(x.abs().max()), float((TS * t).abs().max()), []
        layers = model._geo_layers()
        for li, m in enumerate(layers):
            W = folded(m)
            op_v = max(op_v, float(W.abs().max()) * (S if li == 0 else 1.0))
            z, tz = x @ W.t() + m.bias, t @ W.t()
            y, g = F.softplus(z, beta=100), torch.sigmoid(100.0 * z)
            ty = g * tz
            out[f"geo_layer{li}_max_abs_activation_log2_units"] = float(S * y.abs().max())
            out[f"geo_layer{li}_max_abs_tangent_operand"] = float(TS * S * ty.abs().max())
            out[f"geo_layer{li}_weight_g_max"] = float(m.weight_g.abs().max())
            if li + 1 < len(layers):                      # (the last hidden layer's activations feed the fp32 head, not a k-loop)
                op_v = max(op_v, float(S * y.abs().max()))
                op_t = max(op_t, float(TS * S * ty.abs().max()))
                t_min_scale.append(float(TS * S * ty.abs().median()))
            x, t = y, ty
        Wd = folded(model.density_linear)
        sdf = x @ Wd.t() + model.density_linear.bias
        dsdf = t @ Wd.t()
        out["max_abs_sdf"], out["max_abs_dsdf_dds"] = float(sdf.abs().max()), float(dsdf.abs().max())
        out["median_tangent_operand_hidden"] = min(t_min_scale) if t_min_scale else 0.0
        # colour network: nabla (|d sdf/d ds| bounds |nabla| up to the unit-ish gradient of ds), view embedding (<= 1), code embedding, relu activations
        view = F.normalize(torch.randn(xyz.shape[0], 3, device=xyz.device, generator=torch.Generator(device=xyz.device).manual_seed(0)), dim=-1)
        parts = ([dsdf.expand(-1, 3)] if model.enable_nablas_input else []) + [d_emb, model.embed_fn_view(view), model.embed_fn_ft(ft)]
        xc = torch.cat(parts, -1)
        op_c = float(xc.abs().max())
        for li, m in enumerate(model._col_layers()):
            op_c = max(op_c, float(m.weight.abs().max()))
            xc = F.relu(xc @ m.weight.t() + m.bias)
            out[f"col_layer{li}_max_abs_activation"] = float(xc.abs().max())
            op_c = max(op_c, float(xc.abs().max()))
        out["max_abs_operand_value_rows"] = max(op_v, op_c)
        out["max_abs_operand_tangent_rows"] = op_t
        chunks = torch.cat([fg.reshape(-1, 4), ft.reshape(-1, 4)], 0).abs().max(-1).values     # the kernels test one chunk of 4 dims at a time;
        out["sincos_fast_range_exceeded_share"] = float((chunks > 1.0e5).float().mean())        # bands = 2: the largest directly evaluated frequency is 1
        out["max_sincos_argument"] = float(chunks.max())
    return out
