"""oracle/render.py -- TEST INFRASTRUCTURE ONLY (CPU parity oracle, numpy fp32).

Restates the reference's volumetric render inner loop:

* cdf_Phi_s / sdf_to_alpha / alpha_to_w        models/renderer.py:13-24, :49-63
* compute_bounded_near_far                     models/renderer.py:66-102
* volume_render -> render_rayschunk            models/renderer.py:105-368 (:162-350)
* get_rays / lift                              utils/rend_util.py:95-118, :123-176
* near_far_from_sphere                         utils/rend_util.py:179-199
* sample_pdf (det=True)                        utils/rend_util.py:276-319

with the kwargs get_model() puts in render_kwargs_test for configs/neumesh_dtu_scan63.yaml
(SURVEY.md section 3.1): perturb=False, bounded_near_far=True, N_samples=N_importance=64,
N_upsample_iters=4, calc_normal per caller.  Only the deterministic (perturb=False) path is
restated: that is the path render.py takes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np

from .field import F32, sigmoid


def torch_linspace01(n: int) -> np.ndarray:
    """torch.linspace(0, 1, n) in fp32, bit-for-bit (checked in tests/test_oracle.py):
    step = 1/(n-1) rounded to fp32; first half step*i, second half fma(-step, n-1-i, 1)."""
    if n == 1:
        return np.zeros(1, F32)
    step = F32(1.0) / F32(n - 1)
    i = np.arange(n)
    lo = np.float64(step) * i
    hi = 1.0 - np.float64(step) * (n - 1 - i)
    return np.where(i < n // 2, lo, hi).astype(F32)


def normalize(x: np.ndarray, eps: float = 1e-12) -> np.ndarray:
    """torch.nn.functional.normalize(x, dim=-1): x / max(||x||_2, eps)."""
    n = np.sqrt(np.sum(x * x, axis=-1, keepdims=True, dtype=F32))
    return (x / np.maximum(n, F32(eps))).astype(F32)


def get_rays(c2w, intrinsics, H: int, W: int):
    """utils/rend_util.py:123-176 (pose-matrix branch, N_rays=-1) with lift() :95-118, all pixels in
    row-major order: x = column, y = row.  c2w [4,4], intrinsics [4,4] -> rays_o, rays_d [H*W,3]."""
    c2w, k = np.asarray(c2w, F32), np.asarray(intrinsics, F32)
    pix = np.arange(H * W)
    x, y = (pix % W).astype(F32), (pix // W).astype(F32)
    fx, fy, cx, cy, sk = k[0, 0], k[1, 1], k[0, 2], k[1, 2], k[0, 1]
    x_lift = ((x - cx + cy * sk / fy - sk * y / fy) / fx).astype(F32)
    y_lift = ((y - cy) / fy).astype(F32)
    d = np.stack([x_lift, y_lift, np.ones_like(x_lift)], -1)
    d = (d / np.sqrt(np.sum(d * d, axis=-1, keepdims=True, dtype=F32))).astype(F32)
    d = np.stack([np.sum(c2w[a, :3] * d, axis=-1, dtype=F32) for a in range(3)], -1).astype(F32)
    return np.broadcast_to(c2w[:3, 3], d.shape).copy(), d


def near_far_from_sphere(rays_o, rays_d, r: float = 1.0):
    """utils/rend_util.py:179-199."""
    dot = np.sum(rays_o * rays_d, axis=-1, keepdims=True, dtype=F32)
    mid = -dot
    near = np.maximum(mid - F32(r), F32(0.0))
    far = np.maximum(mid + F32(r), F32(r))
    return near.astype(F32), far.astype(F32)


def sdf_to_alpha(sdf, s):
    """models/renderer.py:17-24."""
    cdf = sigmoid(sdf * s)
    alpha = (cdf[..., :-1] - cdf[..., 1:]) / (cdf[..., :-1] + F32(1e-10))
    return cdf, np.maximum(alpha, F32(0.0)).astype(F32)


def alpha_to_w(alpha):
    """models/renderer.py:49-63: alpha * cumprod([1, 1-alpha+1e-10])[:-1].

    torch.cumprod / torch.cumsum on CPU run sequentially along the last dim with a float64
    accumulator (at::acc_type<float, /*is_cuda=*/false>) and round each output to fp32; that
    is what is restated here (verified against the reference by gen_golden.py)."""
    shifted = np.concatenate([np.ones(alpha.shape[:-1] + (1,), F32), F32(1.0) - alpha + F32(1e-10)], axis=-1)
    T = np.cumprod(shifted.astype(np.float64), axis=-1).astype(F32)
    return (alpha * T[..., :-1]).astype(F32)


def sample_pdf_det(bins, weights, n_importance: int, eps: float = 1e-5, u=None):
    """utils/rend_util.py:276-319 with det=True; u [...,n_importance] given = det=False with the
    caller's uniform randoms in place of torch.rand (:293-296)."""
    weights = weights + F32(1e-5)
    pdf = weights / np.sum(weights, axis=-1, keepdims=True, dtype=F32)
    cdf = np.empty(pdf.shape[:-1] + (pdf.shape[-1] + 1,), F32)
    cdf[..., 0] = 0
    cdf[..., 1:] = np.cumsum(pdf.astype(np.float64), axis=-1).astype(F32)  # float64 accumulator, see alpha_to_w
    flat_cdf = cdf.reshape(-1, cdf.shape[-1])
    if u is None:
        u = torch_linspace01(n_importance)
        inds = np.stack([np.searchsorted(row, u, side="left") for row in flat_cdf])
    else:
        u = np.ascontiguousarray(u, dtype=F32)
        inds = np.stack([np.searchsorted(row, ur, side="left") for row, ur in zip(flat_cdf, u.reshape(-1, n_importance))])
    inds = inds.reshape(cdf.shape[:-1] + (n_importance,))
    below = np.maximum(inds - 1, 0)
    above = np.minimum(inds, cdf.shape[-1] - 1)
    cdf_b = np.take_along_axis(cdf, below, axis=-1)
    cdf_a = np.take_along_axis(cdf, above, axis=-1)
    bin_b = np.take_along_axis(bins, below, axis=-1)
    bin_a = np.take_along_axis(bins, above, axis=-1)
    denom = cdf_a - cdf_b
    denom = np.where(denom < F32(eps), F32(1.0), denom)
    t = (u - cdf_b) / denom
    return (bin_b + t * (bin_a - bin_b)).astype(F32)


def compute_bounded_near_far(field, rays_o, rays_d, near, far, sample_grid: int = 256, distance_thresh: float = 0.1):
    """models/renderer.py:66-102.  rays_*: [R,3], near/far: [R,1]."""
    t = torch_linspace01(sample_grid)
    d = (near * (F32(1.0) - t) + far * t).astype(F32)          # [R,G]
    pts = (rays_o[:, None, :] + d[..., None] * rays_d[:, None, :]).astype(F32)
    ds, _, _ = field.compute_distance(pts)
    mask = ds[..., 0] < F32(distance_thresh)
    mf = mask.astype(F32)
    nf = (~mask).astype(F32)
    n2 = np.min(d * mf + nf * F32(1e10), axis=-1, keepdims=True)
    n2 = np.where(n2 > F32(1e5), near, n2)
    f2 = np.max(d * mf - nf * F32(1e10), axis=-1, keepdims=True)
    f2 = np.where(f2 < F32(-1e5), far, f2)
    too_close = (f2 - n2) < F32(0.1)
    f2 = np.where(too_close, f2 + F32(0.05), f2)
    n2 = np.where(too_close, n2 - F32(0.05), n2)
    return n2.astype(F32), f2.astype(F32), ds[..., 0]


@dataclass
class RenderConfig:
    """Subset of volume_render's signature (models/renderer.py:105-135) that changes results."""
    obj_bounding_radius: float = 1.0
    N_samples: int = 64
    N_importance: int = 64
    N_upsample_iters: int = 4
    bounded_near_far: bool = True
    calc_normal: bool = True
    white_bkgd: bool = False
    near_bypass: float | None = None   # models/renderer.py:171-174: replace every ray's near / far
    far_bypass: float | None = None


def upsample_step(d, sdf, it: int, n_new: int, u=None):
    """One iteration of the up-sampling loop, models/renderer.py:209-245 (weights -> d_fine)."""
    prev_sdf, next_sdf = sdf[..., :-1], sdf[..., 1:]
    prev_z, next_z = d[..., :-1], d[..., 1:]
    mid_sdf = (prev_sdf + next_sdf) * F32(0.5)
    dot = (next_sdf - prev_sdf) / (next_z - prev_z + F32(1e-5))
    prev_dot = np.concatenate([np.zeros_like(dot[..., :1]), dot[..., :-1]], axis=-1)
    dot = np.clip(np.minimum(prev_dot, dot), F32(-10.0), F32(0.0))
    dist = next_z - prev_z
    prev_esti = mid_sdf - dot * dist * F32(0.5)
    next_esti = mid_sdf + dot * dist * F32(0.5)
    s = F32(256 * (2 ** it))
    prev_cdf = sigmoid(prev_esti * s)
    next_cdf = sigmoid(next_esti * s)
    alpha = (prev_cdf - next_cdf + F32(1e-5)) / (prev_cdf + F32(1e-5))
    w = alpha_to_w(alpha.astype(F32))
    return sample_pdf_det(d, w, n_new, u=u), w


def render_rays(field, rays_o, rays_d, cfg: RenderConfig = RenderConfig(), detailed: bool = False, u_rand=None) -> Dict[str, np.ndarray]:
    """models/renderer.py:150-153 + render_rayschunk :162-350 for one chunk of R rays (B squeezed).
    u_rand (optional): one [R, N_importance / N_upsample_iters] array of uniform numbers per up-sampling iteration = perturb=True with the
    caller's numbers in place of torch.rand (sample_pdf(det=False), utils/rend_util.py:298-302)."""
    rays_o = np.ascontiguousarray(rays_o, dtype=F32).reshape(-1, 3)
    rays_d = normalize(np.ascontiguousarray(rays_d, dtype=F32).reshape(-1, 3))  # :153
    near, far = near_far_from_sphere(rays_o, rays_d, cfg.obj_bounding_radius)
    out: Dict[str, np.ndarray] = {}
    if cfg.bounded_near_far:
        out["near_sphere"], out["far_sphere"] = near, far
        near, far, probe_ds = compute_bounded_near_far(field, rays_o, rays_d, near, far)
        if detailed:
            out["probe_ds"] = probe_ds
    if cfg.near_bypass is not None:                                          # :171-172
        near = (F32(cfg.near_bypass) * np.ones_like(near)).astype(F32)
    if cfg.far_bypass is not None:                                           # :173-174
        far = (F32(cfg.far_bypass) * np.ones_like(far)).astype(F32)
    out["near"], out["far"] = near, far

    t = torch_linspace01(cfg.N_samples)
    d = (near * (F32(1.0) - t) + far * t).astype(F32)                       # :193-194
    pts = (rays_o[:, None, :] + d[..., None] * rays_d[:, None, :]).astype(F32)
    sdf = field.forward_density_only(pts)[..., 0]                            # :202-207
    out["d_coarse"], out["sdf_coarse"] = d, sdf
    n_new = cfg.N_importance // cfg.N_upsample_iters
    for it in range(cfg.N_upsample_iters if n_new > 0 else 0):               # :208-258 (no new samples: every iteration cats nothing and re-sorts sorted depths)
        d_fine, _ = upsample_step(d, sdf, it, n_new, u=None if u_rand is None else u_rand[it])
        pts_f = (rays_o[:, None, :] + d_fine[..., None] * rays_d[:, None, :]).astype(F32)
        sdf_f = field.forward_density_only(pts_f)[..., 0]
        d = np.concatenate([d, d_fine], axis=-1)
        sdf = np.concatenate([sdf, sdf_f], axis=-1)
        order = np.argsort(d, axis=-1, kind="stable")
        d = np.take_along_axis(d, order, axis=-1)
        sdf = np.take_along_axis(sdf, order, axis=-1)
        if detailed:
            out[f"d_fine_{it}"] = d_fine
    out["d_all"] = d
    out.update(render_at_depths(field, rays_o, rays_d, d, cfg, detailed, normalized=True))
    return out


def render_at_depths(field, rays_o, rays_d, d_all, cfg: RenderConfig = RenderConfig(), detailed: bool = False,
                     normalized: bool = False) -> Dict[str, np.ndarray]:
    """The tail of render_rayschunk (models/renderer.py:264-333) on GIVEN sorted sample depths d_all [R,N]:
    field + nablas at the samples, radiance at the mid-points, alpha / weights / compositing.  (Used with the
    reference's own depths this compares everything after the sample placement without the placement's chaos.)"""
    rays_o = np.ascontiguousarray(rays_o, dtype=F32).reshape(-1, 3)
    rays_d = np.ascontiguousarray(rays_d, dtype=F32).reshape(-1, 3)
    if not normalized:
        rays_d = normalize(rays_d)
    d_all = np.ascontiguousarray(d_all, dtype=F32)
    out: Dict[str, np.ndarray] = {}

    pts = (rays_o[:, None, :] + rays_d[:, None, :] * d_all[..., None]).astype(F32)      # :264
    d_mid = (F32(0.5) * (d_all[..., 1:] + d_all[..., :-1])).astype(F32)                 # :266
    pts_mid = (rays_o[:, None, :] + rays_d[:, None, :] * d_mid[..., None]).astype(F32)  # :267
    if cfg.calc_normal:
        sdf, nablas = field.forward_with_nablas(pts)                                    # :271-272
    else:
        sdf, nablas = field.forward_density_only(pts), None
    sdf = sdf[..., 0]
    cdf, alpha = sdf_to_alpha(sdf, field.forward_s())                                   # :278
    view = np.broadcast_to(rays_d[:, None, :], pts_mid.shape)
    sdf_mid, radiance, nablas_mid = field.forward(pts_mid, view)                        # :279-282
    w = alpha_to_w(alpha)                                                               # :302
    rgb = np.sum(w[..., None] * radiance, axis=-2, dtype=F32)                           # :304
    depth = np.sum(w / (np.sum(w, axis=-1, keepdims=True, dtype=F32) + F32(1e-10)) * d_mid, axis=-1, dtype=F32)
    acc = np.sum(w, axis=-1, dtype=F32)                                                 # :313
    if cfg.white_bkgd:
        rgb = rgb + (F32(1.0) - acc[..., None])                                         # :315-316
    out.update(rgb=rgb.astype(F32), depth_volume=depth.astype(F32), mask_volume=acc.astype(F32))
    if cfg.calc_normal:
        nmap = normalize(nablas)
        n_pts = min(w.shape[-1], nmap.shape[-2])
        out["normals_volume"] = np.sum(nmap[..., :n_pts, :] * w[..., :n_pts, None], axis=-2, dtype=F32)  # :327-333
    if detailed:
        out.update(implicit_surface=sdf, radiance=radiance, alpha=alpha, cdf=cdf,
                   visibility_weights=w, d_final=d_mid, sdf_mid=sdf_mid[..., 0], nablas_mid=nablas_mid)
        if cfg.calc_normal:
            out["implicit_nablas"] = nablas
    return out
