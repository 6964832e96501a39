"""oracle/compare.py -- TEST INFRASTRUCTURE ONLY: comparators used by tests/ and gen_golden.py.

Why per-sample arrays cannot be compared index-by-index
-------------------------------------------------------
`sample_pdf(det=True)` (utils/rend_util.py:276-319) draws u = linspace(0,1,16); its last sample
(u = 1.0) is placed by `searchsorted(cdf, 1.0)`, i.e. by whether the fp32 running sum
`cdf[-1]` rounded to >= 1.0 or to < 1.0, and -- because the last bin's pdf is < eps=1e-5 for
practically every ray, which triggers `denom[denom < eps] = 1` -- the sample lands either ON
`bins[-1]`, ON `bins[-2]`, or (pdf of the last bin just above eps) a fraction
(cdf[-1]-1)/pdf_last before `bins[-1]`.  All of these lie inside the last, ~zero-weight bin, so the
rendered pixel does not change (measured 2e-7 on RGB), but which duplicate appears depends on
the summation order of `torch.sum`/`torch.cumsum` (CPU: float64 accumulator, sequential; CUDA:
fp32 parallel scan).  The reference itself is therefore not reproducible across its own
back ends at the level of sample *indices*; what is reproducible is the SET of distinct
sample positions per ray, the field values at those positions, and the composited pixel.

Second-order effect (measured by gen_golden.py, oracle vs the reference's own CPU run): the
duplicate that was placed differently carries a different alpha in the next up-sampling
iteration, which moves that iteration's cdf by ~1e-5 and hence samples in very-low-pdf bins
by up to a few 1e-4 in depth (a small fraction of all samples).  Those samples carry ~zero
weight; pixels agree to ~2e-7.  So the gates are: composited rgb/depth/acc/normals <= 1e-4,
field values equal (<= 1e-5) on all position-matched samples, and >= 90 % of samples
position-matched to 2e-6.
"""
from __future__ import annotations

import numpy as np


def depth_set_distance(d_got, d_want):
    """Symmetric per-ray set distance between two sample-position lists: for every sample of
    one side the distance to the nearest sample of the other.  Returns (max, fraction of
    samples farther than 2e-6 from any partner)."""
    d_got = np.asarray(d_got, np.float64)
    d_want = np.asarray(d_want, np.float64)
    worst, far, total = 0.0, 0, 0
    for g, w in zip(d_got.reshape(-1, d_got.shape[-1]), d_want.reshape(-1, d_want.shape[-1])):
        ws, gs = np.sort(w), np.sort(g)
        for a, b in ((g, ws), (w, gs)):
            j = np.clip(np.searchsorted(b, a), 1, len(b) - 1)
            dist = np.minimum(np.abs(a - b[j - 1]), np.abs(a - b[j]))
            worst = max(worst, float(dist.max()))
            far += int(np.sum(dist > 2e-6))
            total += dist.size
    return worst, far / max(total, 1)


def max_err_matched_by_depth(d_got, v_got, d_want, v_want, tol_d=2e-6):
    """Compare point quantities (sdf, radiance, nabla at sample j) after matching each `got`
    sample to the nearest `want` sample of the same ray; only pairs closer than tol_d count.
    Returns (max |v_got - v_want| over matched pairs, fraction of `got` samples matched)."""
    d_got = np.asarray(d_got, np.float64)
    d_want = np.asarray(d_want, np.float64)
    v_got = np.asarray(v_got, np.float64)
    v_want = np.asarray(v_want, np.float64)
    if v_got.ndim == d_got.ndim:
        v_got, v_want = v_got[..., None], v_want[..., None]
    R = d_got.reshape(-1, d_got.shape[-1]).shape[0]
    dg, dw = d_got.reshape(R, -1), d_want.reshape(R, -1)
    vg, vw = v_got.reshape(R, dg.shape[1], -1), v_want.reshape(R, dw.shape[1], -1)
    worst, matched = 0.0, 0
    for r in range(R):
        order = np.argsort(dw[r], kind="stable")
        ws = dw[r][order]
        j = np.clip(np.searchsorted(ws, dg[r]), 1, len(ws) - 1)
        pick = np.where(np.abs(dg[r] - ws[j - 1]) <= np.abs(dg[r] - ws[j]), j - 1, j)
        m = np.abs(dg[r] - ws[pick]) <= tol_d
        matched += int(m.sum())
        if m.any():
            worst = max(worst, float(np.max(np.abs(vg[r][m] - vw[r][order[pick]][m]))))
    return worst, matched / dg.size


def max_err_on_identical_points(p_got, v_got, p_want, v_want):
    """Compare point quantities on the samples whose xyz is BIT-identical on both sides (same
    ray).  The NeuMesh field is discontinuous where the K-NN set changes, so even a 1-ulp
    shift of a sample may legitimately jump the SDF by ~1e-3; identical points must agree.
    p_*: [R,N,3] float32, v_*: [R,N] or [R,N,C].  Returns (max err, fraction identical)."""
    p_got = np.ascontiguousarray(p_got, np.float32)
    p_want = np.ascontiguousarray(p_want, np.float32)
    v_got = np.asarray(v_got, np.float64).reshape(p_got.shape[0], p_got.shape[1], -1)
    v_want = np.asarray(v_want, np.float64).reshape(p_want.shape[0], p_want.shape[1], -1)
    worst, hit = 0.0, 0
    for r in range(p_got.shape[0]):
        lut = {}
        for j, key in enumerate(p_want[r].view(np.uint32).reshape(-1, 3)):
            lut.setdefault(key.tobytes(), j)
        for i, key in enumerate(p_got[r].view(np.uint32).reshape(-1, 3)):
            j = lut.get(key.tobytes())
            if j is not None:
                hit += 1
                worst = max(worst, float(np.max(np.abs(v_got[r, i] - v_want[r, j]))))
    return worst, hit / (p_got.shape[0] * p_got.shape[1])


def psnr(a, b):
    """utils/metric_util.py:6-16 of the reference: -10 log10(mean((a-b)^2))."""
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return float("inf") if mse == 0 else -10.0 * np.log10(mse)


def first_divergent_stage(got: dict, want: dict, ray: int):
    """Diagnostics for a ray whose pixel differs: the first stage of render_rayschunk
    (models/renderer.py:162-259) whose output differs from the reference's by more than the last bit.
    got / want: {"near_far": [R,2], "sdf_coarse": [R,Ns], "d_iter1".."d_iterK": [R,n_k] sorted depths}.
    Returns (stage name, max difference in units of the last place, number of differing entries)."""
    stages = ["near_far", "sdf_coarse"] + sorted(k for k in want if k.startswith("d_iter"))
    for name in stages:
        if name not in got or name not in want:
            continue
        a = np.asarray(got[name][ray], np.float32).reshape(-1)
        b = np.asarray(want[name][ray], np.float32).reshape(-1)
        ulp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
        diff = np.abs(a.astype(np.float64) - b.astype(np.float64)) / np.maximum(ulp, 1e-45)
        # sdf values are outputs of a 3-layer MLP: "the last bit" of the field is its 3e-6 parity bound
        if name == "sdf_coarse":
            diff = np.abs(a.astype(np.float64) - b.astype(np.float64)) / 3e-6
        bad = diff > 1.0
        if bad.any():
            return name, float(diff.max()), int(bad.sum())
    return None, 0.0, 0
