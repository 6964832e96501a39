"""Synthetic scenes for benchmarking and parity tests (numpy only, no device code).

There is no DTU / NeRF-synthetic data, checkpoint or prior-mesh ``.ply`` in the build
environment, so "DTU scan63" in BASELINE.json is honoured in *shape* only (SURVEY.md section 8d,
scene **S-DTU**): a Fibonacci-sphere mesh with V ~ 1.4e5 vertices at the vertex spacing of a
256^3 marching-cubes mesh (reference: extract_mesh.py:40-45,139), 32-d N(0,1) geometry /
colour codes (reference: models/frameworks/neumesh/neumesh.py:47-52), indicator vectors =
normals + noise, a pin-hole camera looking at the origin.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class SyntheticMesh:
    vertices: np.ndarray        # [V,3] float32
    vertex_normals: np.ndarray  # [V,3] float32 (analytic, unit)

    @property
    def num_vertices(self) -> int:
        return int(self.vertices.shape[0])


def fibonacci_blob(V: int = 140_000, radius: float = 0.75, bump: float = 0.05) -> SyntheticMesh:
    """Vertices on r(theta,phi) = radius + bump*sin(7 theta)*cos(5 phi), Fibonacci-spiral sampled."""
    i = np.arange(V, dtype=np.float64) + 0.5
    z = 1.0 - 2.0 * i / V
    theta = np.arccos(np.clip(z, -1.0, 1.0))
    phi = (np.pi * (1.0 + 5.0 ** 0.5) * i) % (2.0 * np.pi)
    st, ct, sp, cp = np.sin(theta), np.cos(theta), np.sin(phi), np.cos(phi)
    r = radius + bump * np.sin(7.0 * theta) * np.cos(5.0 * phi)
    rhat = np.stack([st * cp, st * sp, ct], -1)
    that = np.stack([ct * cp, ct * sp, -st], -1)
    phat = np.stack([-sp, cp, np.zeros_like(sp)], -1)
    dr_dt = bump * 7.0 * np.cos(7.0 * theta) * np.cos(5.0 * phi)
    dr_dp = -bump * 5.0 * np.sin(7.0 * theta) * np.sin(5.0 * phi)
    n = rhat - (dr_dt / r)[:, None] * that - (dr_dp / (r * np.maximum(st, 1e-6)))[:, None] * phat
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    return SyntheticMesh((rhat * r[:, None]).astype(np.float32), n.astype(np.float32))


def random_codes(V: int, dim: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).standard_normal((V, dim)).astype(np.float32)


def noisy_indicator(normals: np.ndarray, seed: int = 3, sigma: float = 0.1) -> np.ndarray:
    rng = np.random.default_rng(seed)
    return (normals + sigma * rng.standard_normal(normals.shape)).astype(np.float32)


def look_at_pose(cam_loc, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """OpenCV-convention camera-to-world 4x4 (x right, y down, z forward)."""
    cam_loc = np.asarray(cam_loc, np.float64)
    fwd = np.asarray(target, np.float64) - cam_loc
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, cam_loc
    return c2w.astype(np.float32)


def orbit_pose(frame: int = 0, n_frames: int = 90, radius: float = 2.2, elevation: float = 0.35) -> np.ndarray:
    a = 2.0 * np.pi * frame / n_frames
    loc = radius * np.array([np.cos(a) * np.cos(elevation), np.sin(a) * np.cos(elevation), np.sin(elevation)])
    return look_at_pose(loc)


def pinhole_intrinsics(H: int, W: int, focal_scale: float = 1.2) -> np.ndarray:
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = K[1, 1] = focal_scale * W
    K[0, 2], K[1, 2] = W / 2.0, H / 2.0
    return K


def camera_rays(c2w: np.ndarray, intrinsics: np.ndarray, H: int, W: int, start: int = 0, count: int = -1):
    """Rays of pixels [start, start+count) in row-major order, the convention of the reference's
    utils/rend_util.py:123-176 (pose-matrix branch, N_rays=-1): pixel (i=col, j=row) is lifted
    to z=1, normalised, rotated by c2w[:3,:3]; origin = c2w[:3,3]."""
    n = H * W if count < 0 else count
    pix = np.arange(start, start + n)
    i = (pix % W).astype(np.float32)
    j = (pix // W).astype(np.float32)
    fx, fy, cx, cy, sk = intrinsics[0, 0], intrinsics[1, 1], intrinsics[0, 2], intrinsics[1, 2], intrinsics[0, 1]
    y = (j - cy) / fy
    x = (i - cx - sk * y) / fx
    d = np.stack([x, y, np.ones_like(x)], -1).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = (d @ c2w[:3, :3].T).astype(np.float32)
    o = np.broadcast_to(c2w[:3, 3].astype(np.float32), d.shape).copy()
    return o, d


def surface_mlp_state(base: dict, shift: float = 0.1, bump_std: float = 0.002, color_gain: float = 12.0,
                      s_value: float = 400.0, speed_factor: float = 10.0, calib=(0.125, -0.095375)) -> dict:
    """MLP weights of a scene WITH a surface, derived deterministically from a default-initialised set
    (`base`: the reference constructor's state dict, e.g. tests/golden/model_seed0.npz).

    A default-initialised geometry MLP is a noise field (sdf < 0 everywhere, every ray opaque). Here
    hidden unit 0 of the three geometry layers carries the projected distance: z = ds + shift stays in
    the near-linear regime of Softplus(beta=100) (100 z from 5 upwards: the curved part and PyTorch's
    threshold-20 linear branch are both crossed close to the surface), the density head reads it back
    with weight 1, and the other 255 units keep their default weights (cut off from unit 0) and add a
    code-driven bump of standard deviation `bump_std` on the surface:  sdf = ds + bump(fg, ds) .
    The colour MLP keeps its weights; only its output layer is multiplied by `color_gain` so that
    the radiance varies with code / view / normal instead of sitting at sigmoid(~0).

    calib = (head_scale, head_bias): bump amplitude and offset, calibrated once on the fibonacci_blob meshes
    (bump mean 0 / -2.7e-4, std 1.9e-3 / 1.6e-3 at V = 3000 / 140 000); None leaves the head of the other
    units unscaled and the bias at -shift.
    weight_norm parametrisation (models/frameworks/neumesh/neumesh.py:76-86,101): weight = g * v / |v|,
    so v = the wanted matrix and g = its row norms (float64 sum, rounded once)."""
    sd = {k: np.array(v, copy=True) for k, v in base.items()}
    f32 = np.float32

    def set_wn(prefix, W):
        W = np.ascontiguousarray(W, dtype=f32)
        sd[prefix + ".weight_v"] = W
        sd[prefix + ".weight_g"] = np.sqrt(np.sum(W.astype(np.float64) ** 2, axis=1, keepdims=True)).astype(f32)

    def eff(prefix):
        v, g = sd[prefix + ".weight_v"].astype(np.float64), sd[prefix + ".weight_g"].astype(np.float64)
        return (g * v / np.sqrt(np.sum(v * v, axis=1, keepdims=True))).astype(f32)

    W0 = eff("pts_linears.0")
    W0[0, :] = 0
    W0[0, 0] = 1.0                       # column 0 of the embedding is the raw ds (models/base.py:59-60)
    set_wn("pts_linears.0", W0)
    sd["pts_linears.0.bias"][0] = f32(shift)
    for name in ("pts_linears.2.0", "pts_linears.3.0"):
        W = eff(name)
        W[0, :] = 0
        W[:, 0] = 0
        W[0, 0] = 1.0
        set_wn(name, W)
        sd[name + ".bias"][0] = 0.0
    head_scale, head_bias = (1.0, -shift) if calib is None else calib
    Wh = eff("density_linear") * f32(head_scale)
    Wh[0, 0] = 1.0
    set_wn("density_linear", Wh)
    sd["density_linear.bias"] = np.array([head_bias], f32)
    sd["color_linear.0.weight"] = (sd["color_linear.0.weight"] * f32(color_gain)).astype(f32)
    sd["ln_s"] = np.array([np.log(s_value) / speed_factor], f32)
    return sd


# ---- texture / geometry editing scenes (fixtures tests/golden/texture_edit_v3000.npz, deform_v3000.npz; bench rows)

def reference_color_state(base: dict, i: int, gain: float = 6.0) -> dict:
    """State dict of the i-th texture REFERENCE model of an editing scene: the main model's weights with its colour network
    (`views_linears.*`, `color_linear.*`) perturbed deterministically and the colour head scaled by `gain`, so that the
    colour it paints is visibly different from the main model's (default-init heads give rgb = 0.5 +- 0.003)."""
    out = dict(base)
    rng = np.random.default_rng(20 + i)
    for k in sorted(base):
        if k.startswith("views_linears"):
            out[k] = (base[k] + 0.05 * (i + 1) * rng.standard_normal(base[k].shape)).astype(np.float32)
        elif k.startswith("color_linear"):
            out[k] = (gain * base[k] + 0.05 * (i + 1) * rng.standard_normal(base[k].shape)).astype(np.float32)
    return out


def edit_scene(vertices: np.ndarray, n_ref: int, rotated: bool, color_dim: int = 32, seed: int = 40):
    """(masks [n_ref, V] bool, edited colour table [V, color_dim] f32, T_r_m list of 4x4 f32 or None): a dozen painted caps per
    reference (contiguous regions, so that points see all-painted, mixed and unpainted neighbour sets), random
    edited codes, and -- if `rotated` -- a rigid transform per reference (editing/texture_neumesh/texture_neumesh.py:21-33)."""
    rng = np.random.default_rng(seed)
    V = vertices.shape[0]
    unit = vertices / np.linalg.norm(vertices, axis=-1, keepdims=True)
    masks = []
    for i in range(n_ref):   # a dozen caps scattered over the object: every view sees painted, mixed and unpainted surface
        centres = rng.standard_normal((12, 3))
        centres /= np.linalg.norm(centres, axis=-1, keepdims=True)
        masks.append((unit @ centres.T).max(-1) > (0.9 if i == 0 else 0.93))
    feats = rng.standard_normal((V, color_dim)).astype(np.float32)
    T_list = None
    if rotated:
        T_list = []
        for i in range(n_ref):
            q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = (q * np.sign(np.linalg.det(q))).astype(np.float32)
            T[:3, 3] = 0.1 * (i + 1)
            T_list.append(T)
    return np.stack(masks), feats, T_list


def deformed_blob(mesh: SyntheticMesh, n_axis: int = 6, twist: float = 0.9):
    """(original mesh with `n_axis` normals snapped to coordinate axes, deformed mesh, snapped indices): the object twisted about z
    by `twist` radians per unit height and stretched 5 % along z, normals carried by the inverse transpose Jacobian (they turn by
    up to ~0.7 rad) -- except that the first three snapped vertices get their normal flipped EXACTLY (cos == -1: the branch of
    render_geometry_editing.py:53,65 that negates the indicator) and the next three keep theirs (rotation vector 0: kornia's
    first-order branch)."""
    v = mesh.vertices.astype(np.float64)
    n0 = mesh.vertex_normals.astype(np.float64).copy()
    snap = np.arange(n_axis) * (mesh.num_vertices // n_axis)
    for j, k in enumerate(snap):
        e = np.zeros(3)
        e[j % 3] = 1.0 if (j // 3) % 2 == 0 else -1.0
        n0[k] = e
    base = SyntheticMesh(mesh.vertices.copy(), n0.astype(np.float32))
    a = twist * v[:, 2]
    c, s_ = np.cos(a), np.sin(a)
    dv = np.stack([c * v[:, 0] - s_ * v[:, 1], s_ * v[:, 0] + c * v[:, 1], 1.05 * v[:, 2]], -1)
    J = np.zeros((v.shape[0], 3, 3))
    J[:, 0, 0], J[:, 0, 1], J[:, 0, 2] = c, -s_, -twist * (s_ * v[:, 0] + c * v[:, 1])
    J[:, 1, 0], J[:, 1, 1], J[:, 1, 2] = s_, c, twist * (c * v[:, 0] - s_ * v[:, 1])
    J[:, 2, 2] = 1.05
    dn = np.einsum("vji,vj->vi", np.linalg.inv(J), n0)        # J^-T n
    dn = dn / np.linalg.norm(dn, axis=-1, keepdims=True)
    dn[snap[:3]] = -n0[snap[:3]]
    dn[snap[3:]] = n0[snap[3:]]
    return base, SyntheticMesh(dv.astype(np.float32), dn.astype(np.float32)), snap
