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
