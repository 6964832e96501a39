"""First-hit surface rendering -- host-side mirror of the reference's ``models/ray_casting.py``
(SURVEY.md section 8 rows a16 / f4): ``root_finding_surface_points`` (:45-200, with ``run_secant_method``
:12-38), ``sphere_tracing_surface_points`` (:203-225) and ``surface_render`` (:228-320), same names,
arguments and return values.

In the reference this module is dead code for NeuMesh (imported nowhere; ``surface_render`` unpacks three
values from ``model.forward`` and reads ``model.implicit_surface``, which only the NeuS teacher framework
provides).  Here it works for both kinds of model: an object with ``implicit_surface`` / a 3-tuple ``forward``
is used exactly as the reference does, a NeuMesh field is queried through ``forward_density_only`` (the SDF) and
its fused ``forward`` (SDF, radiance and nabla at the hit points in one HIP call).

For a NeuMesh field the whole root finding is ONE C call, ``nm_surface_hits`` (csrc/nm_surface.h): the proposals are walked
in blocks of 16 per ray over the compacted list of rays that have not met their first sign change yet (K-NN + distance +
code gather, then the geometry MLP), the secant steps run on the list of hits, all bookkeeping in kernels with the
reference's own arithmetic -- depths and masks agree with the reference value for value.  Any other ``surface_query_fn``
(a NeuS teacher, a wrapper model) takes the torch-op form below, which queries the function it is given in the same
blocks.  Everything is inference (torch.no_grad), as in the reference.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Union

import numpy as np
import torch
import torch.nn.functional as F


def run_secant_method(f_low, f_high, d_low, d_high, rays_o_masked, rays_d_masked, implicit_surface_query_fn, n_secant_steps,
                      logit_tau):
    """models/ray_casting.py:12-38: regula falsi between the last proposal outside (f_high > 0) and the first one
    inside (f_low < 0) the surface; modifies the four bracket tensors in place like the reference."""
    def estimate():
        return -f_low * (d_high - d_low) / (f_high - f_low) + d_low

    d_pred = estimate()
    for _ in range(n_secant_steps):
        p_mid = rays_o_masked + d_pred.unsqueeze(-1) * rays_d_masked
        with torch.no_grad():
            f_mid = implicit_surface_query_fn(p_mid).squeeze(-1) - logit_tau
        inside = f_mid < 0
        # (unconditional masked writes: the reference's `if ind.sum() > 0` guards only skip empty assignments)
        d_low[inside], f_low[inside] = d_pred[inside], f_mid[inside]
        d_high[~inside], f_high[~inside] = d_pred[~inside], f_mid[~inside]
        d_pred = estimate()
    return d_pred


_EARLY_BLOCK = 32


def _native_root_finding(surface, rays_o, rays_d, near, far, N_steps, logit_tau, method, N_secant_steps, fill_inf):
    """nm_surface_hits for a _NeuMeshSurface (flattened rays [M,3]; near / far: floats or [M] tensors)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    model = surface.model
    dev = rays_o.device
    ro, rd = rays_o.reshape(-1, 3).float().contiguous(), rays_d.reshape(-1, 3).float().contiguous()
    M = ro.shape[0]
    cfg = _lib.SurfaceCfg()
    cfg.N_steps, cfg.logit_tau, cfg.fill_inf, cfg.scene_radius = int(N_steps), float(logit_tau), int(bool(fill_inf)), 1.0
    cfg.n_secant_steps = int(N_secant_steps) if method == "secant" else -1
    nf = None
    if isinstance(near, torch.Tensor) or isinstance(far, torch.Tensor):
        ones = torch.ones(M, device=dev)
        nf = torch.stack([(near.reshape(-1).float() if isinstance(near, torch.Tensor) else near * ones),
                          (far.reshape(-1).float() if isinstance(far, torch.Tensor) else far * ones)], -1).contiguous()
    else:
        cfg.near, cfg.far = float(near), float(far)
    d = torch.empty(M, device=dev)
    pt = torch.empty((M, 3), device=dev)
    mask = torch.empty(M, dtype=torch.uint8, device=dev)
    msc = torch.empty(M, dtype=torch.uint8, device=dev)
    field = model.field_handle()
    ws = torch.empty(max(int(lib.nm_surface_workspace_bytes(field, max(M, 1))), 256), dtype=torch.uint8, device=dev)
    t, keep = model.field_tables()
    with torch.cuda.device(dev):
        for _attempt in range(2):
            _lib.check(lib.nm_surface_hits(field, model.grid_for(dev).grid.handle, C.byref(t), _lib.ptr(ro), _lib.ptr(rd), M, _lib.ptr(nf), C.byref(cfg),
                                           _lib.ptr(d), _lib.ptr(pt), _lib.ptr(mask), _lib.ptr(msc), _lib.ptr(ws), _lib.current_stream(dev)), "nm_surface_hits")
            if model.check_fp16_range():
                break
            field = model.field_handle()
    del keep
    return d, pt, mask.bool(), msc.bool()


def _proposal_values_until_first_sign_change(surface_query_fn, rays_o, rays_d, d_proposal, logit_tau):
    """The [B, N_rays, N_steps] proposal values root_finding_surface_points reads -- evaluated only as far as each ray's
    first sign change.  Everything the routine returns depends on the proposal values through the first sign change alone
    (the first proposal's sign, the index of the first negative product val_j * val_j+1, the two values and depths around
    it), so the proposals beyond it need not be evaluated: they are left at +1 here (the products after the first change are
    never looked at: the cost minimum of :106-117 is already taken by the first one).  Rays are walked in blocks of _EARLY_BLOCK proposals; a ray leaves
    the walk after the block that holds its first sign change.  Same outputs as evaluating all N_steps proposals."""
    B, R, N = d_proposal.shape
    val = torch.ones((B * R, N), device=d_proposal.device, dtype=torch.float32)
    ro, rd, dp = rays_o.reshape(B * R, 3), rays_d.reshape(B * R, 3), d_proposal.reshape(B * R, N)
    alive = torch.arange(B * R, device=d_proposal.device)
    start = 0
    while alive.numel() > 0:
        stop = min(N, start + _EARLY_BLOCK + 1)   # one proposal of overlap: the product at the block's last index needs its successor
        cols = slice(start if start == 0 else start + 1, stop)   # (the first column of a later block was evaluated as the previous block's overlap)
        if cols.start < cols.stop:
            pts = ro[alive].unsqueeze(-2) + dp[alive, cols].unsqueeze(-1) * rd[alive].unsqueeze(-2)
            val[alive, cols] = (surface_query_fn(pts.unsqueeze(0)).reshape(alive.numel(), -1) - logit_tau).float()
        seg = val[alive, start:stop]
        changed = (seg[:, :-1] * seg[:, 1:] < 0).any(dim=-1)
        alive = alive[~changed]
        if stop >= N:
            break
        start = stop - 1
    # rays that left early: everything after their first sign change must not produce an EARLIER-ranked cost; +1 fill gives
    # products of sign(val_last_evaluated) with +1 and +1 * +1 -- if the last evaluated value is negative that is one more
    # negative product, but at a later index (lower cost magnitude), so the minimum stays at the first change.
    return val.reshape(B, R, N)


def root_finding_surface_points(surface_query_fn, rays_o: torch.Tensor, rays_d: torch.Tensor,
                                near: Union[float, torch.Tensor] = 0.0, far: Union[float, torch.Tensor] = 6.0,
                                batched=True, batched_info={}, N_steps=256, logit_tau=0.0, method="secant", N_secant_steps=8,
                                fill_inf=True, early_exit=True):
    """models/ray_casting.py:45-200.  rays_o / rays_d: [(B), N_rays, 3] (rays_d normalised); near / far: float or
    [(B), N_rays].  Returns (d_pred_out [(B),N_rays], pt_pred [(B),N_rays,3], mask, mask_sign_change).
    Sign convention: surface value > 0 outside, < 0 inside; a hit is the FIRST sign change along the ray, and it
    must go from outside to inside with the ray's first proposal outside."""
    with torch.no_grad():
        device = rays_o.device
        if not batched:
            rays_o, rays_d = rays_o.unsqueeze(0), rays_d.unsqueeze(0)
            near = near.unsqueeze(0) if isinstance(near, torch.Tensor) else near
            far = far.unsqueeze(0) if isinstance(far, torch.Tensor) else far
        B, N_rays = rays_o.shape[0], rays_o.shape[-2]
        native = (early_exit and isinstance(surface_query_fn, _NeuMeshSurface) and hasattr(surface_query_fn.model, "field_handle")
                  and getattr(surface_query_fn.model, "fused_supported", lambda: True)()
                  and rays_o.is_cuda and not os.environ.get("NEUMESH_NO_SURFACE_KERNEL"))
        if native:   # the whole routine as one C call (nm_surface_hits); same outputs as the torch-op form below
            d_pred_out, pt_pred, mask, mask_sign_change = (x.reshape(B, N_rays, *x.shape[1:]) for x in _native_root_finding(
                surface_query_fn, rays_o, rays_d, near, far, N_steps, logit_tau, method, N_secant_steps, fill_inf))
            if not batched:
                d_pred_out, pt_pred, mask, mask_sign_change = d_pred_out[0], pt_pred[0], mask[0], mask_sign_change[0]
            return d_pred_out, pt_pred, mask, mask_sign_change
        t = torch.linspace(0.0, 1.0, N_steps, device=device)[None, None, :]
        if not isinstance(near, torch.Tensor):
            near = near * torch.ones(rays_o.shape[:-1], device=device)
        if not isinstance(far, torch.Tensor):
            far = far * torch.ones(rays_o.shape[:-1], device=device)
        d_proposal = near[..., None] * (1 - t) + far[..., None] * t                              # [B, N_rays, N_steps]
        if early_exit and N_steps > 2 * _EARLY_BLOCK:
            val = _proposal_values_until_first_sign_change(surface_query_fn, rays_o, rays_d, d_proposal, logit_tau)
        else:
            val = surface_query_fn(rays_o.unsqueeze(-2) + d_proposal.unsqueeze(-1) * rays_d.unsqueeze(-2)) - logit_tau
        mask_0_not_occupied = val[..., 0] > 0
        # cost = sign(val_j * val_j+1) * (N_steps - j): its minimum is the first sign change (:106-117)
        sign_matrix = torch.cat([torch.sign(val[..., :-1] * val[..., 1:]), torch.ones([B, N_rays, 1], device=device)], dim=-1)
        cost_matrix = sign_matrix * torch.arange(N_steps, 0, -1, device=device).float()
        values, indices = torch.min(cost_matrix, -1)
        mask_sign_change = values < 0
        mask_pos_to_neg = torch.gather(val, -1, indices.unsqueeze(-1)).squeeze(-1) > 0
        mask = mask_sign_change & mask_pos_to_neg & mask_0_not_occupied

        def at(x, idx):  # x[..., idx] per ray, hit rays only
            return torch.gather(x, -1, idx.unsqueeze(-1)).squeeze(-1)[mask]

        d_high, f_high = at(d_proposal, indices), at(val, indices)
        nxt = torch.clamp(indices + 1, max=N_steps - 1)
        d_low, f_low = at(d_proposal, nxt), at(val, nxt)
        rays_o_masked, rays_d_masked = rays_o[mask], rays_d[mask]
        if method == "secant" and bool(mask.any()):
            d_pred = run_secant_method(f_low, f_high, d_low, d_high, rays_o_masked, rays_d_masked, surface_query_fn,
                                       N_secant_steps, logit_tau)
        else:
            d_pred = torch.ones(rays_o_masked.shape[0], device=device)
        pt_pred = torch.ones([B, N_rays, 3], device=device)
        pt_pred[mask] = rays_o_masked + d_pred.unsqueeze(-1) * rays_d_masked
        d_pred_out = torch.ones([B, N_rays], device=device)
        d_pred_out[mask] = d_pred
        # no hit (no sign change, inside-to-outside first, or the first proposal already inside): inf or `far`
        d_pred_out[~mask] = np.inf if fill_inf else far[~mask]
        d_pred_out[~mask_0_not_occupied] = 0          # the ray starts inside: depth 0
        if not batched:
            d_pred_out, pt_pred, mask, mask_sign_change = d_pred_out[0], pt_pred[0], mask[0], mask_sign_change[0]
    return d_pred_out, pt_pred, mask, mask_sign_change


def sphere_tracing_surface_points(implicit_surface, rays_o, rays_d, near=0.0, far=6.0, batched=True, batched_info={}, N_iters=20):
    """models/ray_casting.py:203-225: d <- d + sdf(o + d * dir), N_iters times; rays leaving [0, far] are frozen."""
    field = implicit_surface.forward if hasattr(implicit_surface, "forward") else implicit_surface
    device = rays_o.device
    d_preds = torch.ones([*rays_o.shape[:-1]], device=device) * near
    mask = torch.ones_like(d_preds, dtype=torch.bool, device=device)
    for _ in range(N_iters):
        surface_val = field(rays_o + rays_d * d_preds[..., :, None])
        d_preds[mask] += surface_val[mask]
        mask[d_preds > far] = False
        mask[d_preds < 0] = False
    return d_preds, rays_o + rays_d * d_preds[..., :, None], mask


class _NeuMeshSurface:
    """What surface_render needs from a model, for a NeuMesh field: the SDF as `implicit_surface` and
    (radiance, sdf, nabla) at the hit points."""

    def __init__(self, model):
        self.model = model

    def forward(self, pts):
        return self.model.forward_density_only(pts).squeeze(-1)

    __call__ = forward

    def shade(self, pts, view_dirs):
        m = self.model
        if hasattr(m, "_fused_forward") and not torch.is_grad_enabled() and m.fused_supported():
            sdf, rgb, nab = m._fused_forward(pts, view_dirs, False)[:3]   # one fused HIP call
            return rgb, sdf, nab
        sdf, nab = m.forward_with_nablas(pts)
        return m.forward(pts, view_dirs)[1], sdf, nab


def surface_render(rays_o: torch.Tensor, rays_d: torch.Tensor, model, calc_normal=True, rayschunk=8192, netchunk=1048576,
                   batched=True, use_view_dirs=True, show_progress=False, ray_casting_algo="", ray_casting_cfgs={},
                   **not_used_kwargs):
    """models/ray_casting.py:228-320: colour / depth / nabla at each ray's first surface hit (black where it misses).
    rays_o / rays_d: [(B,) N_rays, 3], rays_d not necessarily normalised.  Returns (colors, depths, extras) with
    extras = {implicit_nablas, mask_surface[, normals_surface]}."""
    with torch.no_grad():
        dim_batchify = 1 if batched else 0
        flat = [rays_d.shape[0], -1, 3] if batched else [-1, 3]
        rays_o = torch.reshape(rays_o, flat).float()
        rays_d = F.normalize(torch.reshape(rays_d, flat).float(), dim=-1)
        if hasattr(model, "implicit_surface"):   # the reference's own model kind (NeuS): forward -> (color, sdf, nablas)
            surface, shade = model.implicit_surface, (lambda p, v: model.forward(p, v))
        else:
            surface = _NeuMeshSurface(model)
            shade = surface.shade

        def render_rayschunk(ro, rd):
            view_dirs = rd if use_view_dirs else None
            if ray_casting_algo == "root_finding":
                d_pred_out, pt_pred, mask, *_ = root_finding_surface_points(surface, ro, rd, batched=batched, **ray_casting_cfgs)
            elif ray_casting_algo == "sphere_tracing":
                d_pred_out, pt_pred, mask = sphere_tracing_surface_points(surface, ro, rd, batched=batched, **ray_casting_cfgs)
            else:
                raise NotImplementedError
            color, _, nablas = shade(pt_pred, view_dirs)
            color = color.clone()
            color[~mask] = 0  # black
            return color.data, d_pred_out.data, nablas.data, mask.data

        rng = range(0, rays_o.shape[dim_batchify], rayschunk)
        if show_progress:
            try:
                from tqdm import tqdm
                rng = tqdm(rng)
            except ImportError:
                pass
        parts = [render_rayschunk(rays_o[:, i:i + rayschunk] if batched else rays_o[i:i + rayschunk],
                                  rays_d[:, i:i + rayschunk] if batched else rays_d[i:i + rayschunk]) for i in rng]
        colors, depths, nablas, masks = (torch.cat([p[k] for p in parts], dim_batchify) for k in range(4))
        extras = OrderedDict([("implicit_nablas", nablas), ("mask_surface", masks)])
        if calc_normal:
            normals = F.normalize(nablas, dim=-1)
            normals[~masks] = 0
            extras["normals_surface"] = normals
        return colors, depths, extras
