"""Volume renderer -- host-side mirror of the reference's ``models/renderer.py``.

``volume_render`` keeps the reference's signature (models/renderer.py:105-135, unknown kwargs
swallowed by ``**dummy_kwargs``) and return value ``(rgb, depth, ret_dict)`` (:368);
``SingleRenderer`` is the same thin nn.Module (:371-377).  For a NeuMesh field under
``torch.no_grad()`` with the deterministic sampler (what render.py uses: perturb=False) each ray
chunk is ONE call of ``nm_render_rays``: ray set-up, 256-probe near/far tightening, coarse
samples, 4x hierarchical up-sampling, SDF/nabla/colour queries and compositing all run as HIP
kernels on the current stream, with no per-stage tensors materialised in Python.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .neumesh import NeuMesh


def cdf_Phi_s(x, s):
    return torch.sigmoid(x * s)


def sdf_to_alpha(sdf, s):
    """models/renderer.py:17-24."""
    cdf = cdf_Phi_s(sdf, s)
    alpha = (cdf[..., :-1] - cdf[..., 1:]) / (cdf[..., :-1] + 1e-10)
    return cdf, torch.clamp_min(alpha, 0)


def alpha_to_w(alpha):
    """models/renderer.py:49-63."""
    ones = torch.ones([*alpha.shape[:-1], 1], device=alpha.device)
    return alpha * torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], dim=-1), dim=-1)[..., :-1]


def make_render_cfg(obj_bounding_radius=1.0, N_samples=64, N_importance=64, N_upsample_iters=4, bounded_near_far=True,
                    calc_normal=False, white_bkgd=False, near_bypass=None, far_bypass=None) -> _lib.RenderCfg:
    c = _lib.RenderCfg()
    c.obj_bounding_radius = float(obj_bounding_radius)
    c.N_samples, c.N_importance, c.N_upsample_iters = int(N_samples), int(N_importance), int(N_upsample_iters)
    c.bounded_near_far, c.calc_normal, c.white_bkgd = int(bool(bounded_near_far)), int(bool(calc_normal)), int(bool(white_bkgd))
    c.probe_grid, c.probe_thresh = 256, 0.1  # compute_bounded_near_far defaults (renderer.py:72-73)
    c.near_bypass = -1.0 if near_bypass is None else float(near_bypass)
    c.far_bypass = -1.0 if far_bypass is None else float(far_bypass)
    return c


class _Workspace:
    """Caller-owned scratch for nm_render_rays, reused across chunks / frames."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes: int, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = None
            self.buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        return self.buf


_WS = _Workspace()


def render_rays_fused(model: NeuMesh, rays_o, rays_d, cfg: _lib.RenderCfg, rayschunk: int, detailed: bool = False,
                      tables=None, progress=None):
    """rays_o / rays_d: [R,3] device tensors.  Returns dict of [R,...] tensors."""
    lib = _lib.load()
    dev = rays_o.device
    if dev.type != "cuda":
        raise _lib.NeuMeshHipError("rays must be on a HIP device (no CPU fallback)")
    rays_o = rays_o.detach().float().reshape(-1, 3).contiguous()
    rays_d = rays_d.detach().float().reshape(-1, 3).contiguous()
    R = rays_o.shape[0]
    N = cfg.N_samples + cfg.N_importance
    out = OrderedDict(rgb=torch.empty((R, 3), device=dev), depth_volume=torch.empty((R,), device=dev),
                      mask_volume=torch.empty((R,), device=dev))
    if cfg.calc_normal:
        out["normals_volume"] = torch.empty((R, 3), device=dev)
    dbg_t = {}
    if detailed:
        dbg_t = dict(d_all=torch.empty((R, N), device=dev), sdf_all=torch.empty((R, N), device=dev),
                     radiance=torch.empty((R, N - 1, 3), device=dev), near_far=torch.empty((R, 2), device=dev))
        if cfg.calc_normal:
            dbg_t["nablas_all"] = torch.empty((R, N, 3), device=dev)
    chunk = max(1, min(int(rayschunk), R))
    ws_bytes = int(lib.nm_render_workspace_bytes(C.byref(cfg), chunk))
    if ws_bytes < 0:
        _lib.check(1, "nm_render_workspace_bytes")
    ws = _WS.get(ws_bytes, dev)
    field, grid = model.field_handle(), model.mesh_grid.grid.handle
    t, keep = tables if tables is not None else model.field_tables()
    with torch.cuda.device(dev):
        stream = _lib.current_stream(dev)
        for i in (range(0, R, chunk) if progress is None else progress(range(0, R, chunk))):
            n = min(chunk, R - i)
            dbg = None
            if detailed:
                dbg = _lib.RenderDebug()
                dbg.near_far = dbg_t["near_far"][i:].data_ptr()
                dbg.d_all = dbg_t["d_all"][i:].data_ptr()
                dbg.sdf_all = dbg_t["sdf_all"][i:].data_ptr()
                dbg.radiance = dbg_t["radiance"][i:].data_ptr()
                dbg.nablas_all = dbg_t["nablas_all"][i:].data_ptr() if cfg.calc_normal else None
                dbg.sdf_coarse = None
            _lib.check(lib.nm_render_rays(
                field, grid, C.byref(t), _lib.ptr(rays_o[i:]), _lib.ptr(rays_d[i:]), n, C.byref(cfg),
                _lib.ptr(out["rgb"][i:]), _lib.ptr(out["depth_volume"][i:]), _lib.ptr(out["mask_volume"][i:]),
                _lib.ptr(out["normals_volume"][i:]) if cfg.calc_normal else None,
                C.byref(dbg) if dbg is not None else None, _lib.ptr(ws), stream), "nm_render_rays")
    del keep
    if detailed:
        s = model.forward_s().detach()
        sdf, d_all = dbg_t["sdf_all"], dbg_t["d_all"]
        cdf, alpha = sdf_to_alpha(sdf, s)
        if cfg.calc_normal:
            out["implicit_nablas"] = dbg_t["nablas_all"]
        out["implicit_surface"] = sdf
        out["radiance"] = dbg_t["radiance"]
        out["alpha"] = alpha
        out["cdf"] = cdf
        out["visibility_weights"] = alpha_to_w(alpha)
        out["d_final"] = 0.5 * (d_all[..., 1:] + d_all[..., :-1])
        out["d_all"] = d_all              # extra (not in the reference's dict)
        out["near_far"] = dbg_t["near_far"]  # extra
    return out


def volume_render(rays_o, rays_d, model, obj_bounding_radius=1.0, batched=False, batched_info={},
                  calc_normal=False, use_view_dirs=True, rayschunk=65536, netchunk=1048576, white_bkgd=False,
                  near_bypass: Optional[float] = None, far_bypass: Optional[float] = None, detailed_output=True,
                  show_progress=False, perturb=False, fixed_s_recp=1 / 64.0, N_samples=64, N_importance=64,
                  N_nograd_samples=2048, N_upsample_iters=4, samples_output=False, bounded_near_far=True,
                  random_color_direction=False, **dummy_kwargs):
    """Same contract as the reference's volume_render (models/renderer.py:105-368)."""
    if batched:
        B = rays_d.shape[0]
        lead = [B, -1]
    else:
        lead = [-1]
    fused_ok = (isinstance(model, NeuMesh) and not torch.is_grad_enabled() and not perturb and not samples_output
                and not random_color_direction and use_view_dirs)
    if not fused_ok:
        raise NotImplementedError(
            "neumesh_amd.volume_render: only the inference path the reference's render.py takes is implemented on the "
            "HIP library (NeuMesh model, torch.no_grad(), perturb=False, no samples_output / random_color_direction). "
            "Training / editing-wrapper renders are the next rows of SURVEY.md section 8(f).")
    cfg = make_render_cfg(obj_bounding_radius, N_samples, N_importance, N_upsample_iters, bounded_near_far, calc_normal,
                          white_bkgd, near_bypass, far_bypass)
    progress = None
    if show_progress:
        try:
            from tqdm import tqdm
            progress = tqdm
        except ImportError:
            progress = None
    flat_o = torch.reshape(rays_o, [-1, 3]).float()
    flat_d = torch.reshape(rays_d, [-1, 3]).float()
    ret = render_rays_fused(model, flat_o, flat_d, cfg, rayschunk, detailed=detailed_output, progress=progress)
    for k in list(ret.keys()):
        v = ret[k]
        ret[k] = v.reshape(*lead, *v.shape[1:]) if batched else v
    return ret["rgb"], ret["depth_volume"], ret


class SingleRenderer(nn.Module):
    """models/renderer.py:371-377."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, rays_o, rays_d, **kwargs):
        return volume_render(rays_o, rays_d, self.model, **kwargs)
