"""oracle/gen_golden.py -- TEST INFRASTRUCTURE ONLY.  Run in the build container:

    python -m oracle.gen_golden            # writes tests/golden/*.npz and prints a report

What it does
------------
1. Imports the REAL reference (/root/reference, read-only) through oracle/refimport and builds
   its NeuMesh model + SingleRenderer on seeded synthetic meshes via the reference's own
   `build_framework` (FRNN replaced by the declared-arithmetic oracle K-NN -- FRNN is an
   external CUDA package that is not part of /root/reference).
2. Runs the reference's own methods (compute_distance / forward_density_only /
   forward_with_nablas / forward / renderer) on seeded inputs.
3. Runs the numpy restatement in oracle/ on the same inputs and REQUIRES agreement
   (tolerances below) -- this is what pins the oracle to the reference.
4. Stores the REFERENCE's outputs as fixtures under tests/golden/ (they travel to the GPU box;
   /root/reference does not).

The fixtures are regenerated only by hand; tests never call this script.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _REPO not in sys.path:
    sys.path.insert(0, _REPO)

from neumesh_amd import synthetic  # noqa: E402
from oracle import compare  # noqa: E402
from oracle import field as ofield  # noqa: E402
from oracle import render as orender  # noqa: E402
from oracle.refimport import harness  # noqa: E402

GOLDEN = os.path.join(_REPO, "tests", "golden")
REPORT = {}


def _check(name, got, want, atol, rtol=0.0, per_point_tol=None):
    """per_point_tol: optional array broadcastable to `want` added to atol element-wise."""
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, (name, got.shape, want.shape)
    diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
    err = float(np.max(diff)) if got.size else 0.0
    scale = float(np.max(np.abs(want))) if want.size else 0.0
    tol = atol + rtol * scale + (0.0 if per_point_tol is None else per_point_tol)
    ok = bool(np.all(diff <= tol))
    REPORT[name] = {"max_abs_err": err, "scale": scale, "atol": atol, "ok": bool(ok)}
    print(f"  {'OK ' if ok else 'BAD'} {name:34s} max|d|={err:.3e} (scale {scale:.3e}, atol {atol:.1e})")
    assert ok, name


def query_points(mesh, n, seed):
    """Near-surface, mid-range, far / outside-bbox points + points exactly on vertices."""
    rng = np.random.default_rng(seed)
    V = mesh.num_vertices
    a = mesh.vertices[rng.integers(0, V, n // 2)] + 0.01 * rng.standard_normal((n // 2, 3))
    b = mesh.vertices[rng.integers(0, V, n // 4)] + 0.15 * rng.standard_normal((n // 4, 3))
    c = rng.uniform(-2.5, 2.5, (n - n // 2 - n // 4 - 16, 3))
    d = mesh.vertices[rng.integers(0, V, 16)]
    return np.concatenate([a, b, c, d]).astype(np.float32)


def oracle_from_reference(model, mesh):
    import torch  # noqa: F401
    cfg = ofield.FieldConfig(speed_factor=float(model.speed_factor),
                             learn_indicator_weight=bool(model.learn_indicator_weight),
                             enable_nablas_input=bool(model.enable_nablas_input))
    return ofield.OracleField(mesh.vertices, {k: v for k, v in model.state_dict().items()}, cfg)


def gen_field_fixture(tag, V, Q, seed, dup=0, mlp_state=None):
    import torch
    print(f"[{tag}] V={V} Q={Q}")
    mesh = synthetic.fibonacci_blob(V)
    if dup:  # duplicated vertices -> exact distance ties, resolved by vertex index
        mesh = synthetic.SyntheticMesh(np.concatenate([mesh.vertices, mesh.vertices[:dup]]),
                                       np.concatenate([mesh.vertex_normals, mesh.vertex_normals[:dup]]))
    model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state)
    orc = oracle_from_reference(model, mesh)
    q = query_points(mesh, Q, seed)
    dirs = orender.normalize(np.random.default_rng(seed + 1).standard_normal((Q, 3)).astype(np.float32))
    tq = torch.from_numpy(q)
    with torch.no_grad():
        r_ds, r_idx, r_w = model.compute_distance(tq)
        r_sdf0 = model.forward_density_only(tq)
    r_sdf, r_nab = model.forward_with_nablas(tq.clone())
    r_sdf2, r_rgb, r_ds2, r_idx2, r_w2 = model.forward(tq.clone(), torch.from_numpy(dirs), return_ds=True)
    r_col = model.forward_color(r_ds2.detach(), torch.from_numpy(dirs), model.color_features,
                                r_idx2, r_w2.detach(), r_nab.detach())
    ref = dict(ds=r_ds, idx=r_idx, w=r_w, sdf=r_sdf0, sdf_wn=r_sdf, nabla=r_nab, rgb=r_rgb, rgb_fc=r_col)
    ref = {k: v.detach().numpy() for k, v in ref.items()}
    # --- oracle restatement vs the reference's own code
    o_ds, o_idx, o_w, o_g = orc.compute_distance(q, want_grad=True)
    assert np.array_equal(o_idx, ref["idx"]), "kNN indices differ (same declared arithmetic!)"
    _check(f"{tag}.ds", o_ds, ref["ds"], 2e-6)
    _check(f"{tag}.w", o_w, ref["w"], 2e-6)
    o_sdf, o_nab = orc.forward_with_nablas(q)
    _check(f"{tag}.sdf", o_sdf, ref["sdf"], 5e-6)
    _check(f"{tag}.sdf(with_nablas)", o_sdf, ref["sdf_wn"], 5e-6)
    # fp32 sensitivity: the d-embedding carries sin/cos(2^7 * ds), so a 1-ulp change of ds moves
    # d sdf/d ds by ~128*ulp(ds)*|.|: tolerance grows with |ds| (render samples have |ds| < ~0.2)
    _check(f"{tag}.nabla(closed form vs autograd)", o_nab, ref["nabla"], 5e-6,
           per_point_tol=2e-4 * np.abs(ref["ds"]))
    o_sdf2, o_rgb, _ = orc.forward(q, dirs)
    _check(f"{tag}.rgb", o_rgb, ref["rgb"], 5e-6)
    _check(f"{tag}.rgb(forward_color)", o_rgb, ref["rgb_fc"], 5e-6)
    np.savez_compressed(
        os.path.join(GOLDEN, f"{tag}.npz"),
        V=np.int64(V), dup=np.int64(dup), q=q, dirs=dirs,
        idx=ref["idx"].astype(np.int32), d2=orc.knn(q)[1],
        ds=ref["ds"], w=ref["w"], dds_dx=o_g, sdf=ref["sdf"], nabla=ref["nabla"], rgb=ref["rgb"],
        s=np.float32(model.forward_s().item()),
    )
    return model


def gen_render_fixture(tag, V, H, W, frame, seed=0, white_bkgd=False, n_samples=64, mlp_state=None):
    import torch
    print(f"[{tag}] V={V} rays={H * W}")
    mesh = synthetic.fibonacci_blob(V)
    model, kw, renderer, _ = harness.build_reference(mesh, seed=seed, mlp_state=mlp_state)
    orc = oracle_from_reference(model, mesh)
    # a patch of a larger virtual image so that the rays actually graze / hit / miss the object
    big = 48
    c2w, Kmat = synthetic.orbit_pose(frame), synthetic.pinhole_intrinsics(big, big, 1.0)
    o_all, d_all_ = synthetic.camera_rays(c2w, Kmat, big, big)
    rows = np.arange(big // 2 - H // 2, big // 2 + H - H // 2)
    cols = np.arange(0, W * 4, 4) % big  # strided columns: centre hits, borders miss
    sel = (rows[:, None] * big + cols[None, :]).reshape(-1)
    rays_o, rays_d = o_all[sel], d_all_[sel] * np.float32(1.7)  # un-normalised on purpose (renderer.py:153)
    kw = dict(kw)
    kw.update(rayschunk=rays_o.shape[0], white_bkgd=white_bkgd, N_samples=n_samples, N_importance=n_samples)
    # the reference does not return d_all; record the points it hands to forward_with_nablas
    # (renderer.py:271-272) and recover d = (p - o) . dir  (exact to ~1e-7, tolerance 2e-6 below)
    seen = {}
    orig_fwn = model.forward_with_nablas

    def spy(xyz):
        seen["pts"] = xyz.detach().clone()
        return orig_fwn(xyz)

    model.forward_with_nablas = spy
    with torch.no_grad():
        rgb, depth, ex = renderer(torch.from_numpy(rays_o)[None], torch.from_numpy(rays_d)[None],
                                  detailed_output=True, **kw)
    model.forward_with_nablas = orig_fwn
    ref = {k: v[0].detach().numpy() for k, v in ex.items()}
    dirn = orender.normalize(rays_d)
    pts = seen["pts"].numpy().reshape(rays_o.shape[0], -1, 3).astype(np.float64)
    ref_d_all = np.sum((pts - rays_o[:, None, :]) * dirn[:, None, :], axis=-1)
    cfg = orender.RenderConfig(white_bkgd=white_bkgd, calc_normal=bool(kw.get("calc_normal", False)),
                               N_samples=n_samples, N_importance=n_samples)
    out = orender.render_rays(orc, rays_o, rays_d, cfg, detailed=True)
    # per-sample arrays: compared as sets / matched by depth (see oracle/compare.py for why)
    worst, far_frac = compare.depth_set_distance(out["d_all"], ref_d_all)
    print(f"    d_all set distance: max {worst:.3e}; samples without a partner within 2e-6: {100 * far_frac:.2f} %")
    REPORT[f"{tag}.d_all"] = {"set_distance_max": worst, "unmatched_fraction": far_frac}
    assert far_frac < 0.10 and worst < 5e-3
    ref_pts = seen["pts"].numpy().reshape(rays_o.shape[0], -1, 3)
    o_pts = (rays_o[:, None, :] + dirn[:, None, :] * out["d_all"][..., None]).astype(np.float32)
    for name, atol in (("implicit_surface", 2e-6), ("implicit_nablas", 2e-5)):
        if name in ref:
            e, frac = compare.max_err_on_identical_points(o_pts, out[name], ref_pts, ref[name])
            _check(f"{tag}.{name}(identical pts {100 * frac:.0f}%)", np.float32(e), np.float32(0.0), atol)
            assert frac > 0.8
    _check(f"{tag}.rgb", out["rgb"], ref["rgb"], 1e-4)
    _check(f"{tag}.depth_volume", out["depth_volume"], ref["depth_volume"], 1e-4)
    _check(f"{tag}.mask_volume", out["mask_volume"], ref["mask_volume"], 1e-4)
    if cfg.calc_normal:
        _check(f"{tag}.normals_volume", out["normals_volume"], ref["normals_volume"], 1e-4)
    acc = ref["mask_volume"]
    print(f"    acc: min {acc.min():.3f} mean {acc.mean():.3f} max {acc.max():.3f}; "
          f"rays on fallback near/far: {int(np.sum(out['near'] == out['near_sphere']))}")
    np.savez_compressed(
        os.path.join(GOLDEN, f"{tag}.npz"),
        V=np.int64(V), rays_o=rays_o, rays_d=rays_d, white_bkgd=np.bool_(white_bkgd),
        N_samples=np.int64(n_samples), calc_normal=np.bool_(cfg.calc_normal),
        near=out["near"], far=out["far"],            # oracle (reference does not expose them)
        d_all=ref_d_all.astype(np.float32),           # reference (recovered from its query points)
        d_coarse=out["d_coarse"], sdf_coarse=out["sdf_coarse"],  # oracle: input of the up-sampling test
        d_final=ref["d_final"], implicit_surface=ref["implicit_surface"], alpha=ref["alpha"],
        radiance=ref["radiance"], visibility_weights=ref["visibility_weights"],
        rgb=ref["rgb"], depth_volume=ref["depth_volume"], mask_volume=ref["mask_volume"],
        normals_volume=ref.get("normals_volume", np.zeros((0, 3), np.float32)),
        implicit_nablas=ref.get("implicit_nablas", np.zeros((0, 3), np.float32)),
    )
    return model


def gen_perturb_fixture(tag="render_v3000_perturb", V=3000, mlp_state=None, seed=91):
    """The stochastic sampler pinned to the reference: renderer(..., perturb=True) draws the stratum positions of sample_pdf(det=False)
    with torch.rand (utils/rend_util.py:298-302; nothing else depends on `perturb`, models/renderer.py:245-247).  The reference is run on the
    render fixture's rays with torch.rand replaced by a recorded sequence of uniform numbers (numpy, seeded); the fixture holds that sequence
    (one [R, 16] block per up-sampling iteration), the sorted depths and the rendered outputs.  The product is fed the same numbers."""
    import torch
    print(f"[{tag}] reference renderer with perturb=True and recorded uniform numbers, V={V}")
    mesh = synthetic.fibonacci_blob(V)
    model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state)
    rf = np.load(os.path.join(GOLDEN, "render_v3000_dtu.npz"))
    ro, rd = torch.from_numpy(rf["rays_o"])[None], torch.from_numpy(rf["rays_d"])[None]
    R = ro.shape[1]
    rng = np.random.default_rng(seed)
    drawn = []
    orig_rand, orig_sort = torch.rand, torch.sort

    def fake_rand(*size, **kwargs):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)) else tuple(size)
        u = rng.random(shape, dtype=np.float32)
        drawn.append(u.reshape(R, -1))
        return torch.from_numpy(u)

    sorted_d = []

    def spy_sort(x, *a, **k):
        out = orig_sort(x, *a, **k)
        sorted_d.append(out[0].detach().clone())
        return out

    kw = dict(kw)
    kw.update(rayschunk=R, calc_normal=True, N_samples=64, N_importance=64, perturb=True, white_bkgd=False)
    torch.rand, torch.sort = fake_rand, spy_sort
    try:
        with torch.no_grad():
            rgb, depth, ex = renderer(ro, rd, detailed_output=True, **kw)
    finally:
        torch.rand, torch.sort = orig_rand, orig_sort
    assert len(drawn) == 4 and all(u.shape == (R, 16) for u in drawn) and len(sorted_d) == 4
    # the oracle's sampler with the same numbers
    orc = oracle_from_reference(model, mesh)
    out = orender.render_rays(orc, rf["rays_o"], rf["rays_d"], orender.RenderConfig(calc_normal=True), detailed=True, u_rand=drawn)
    same = np.all(out["d_all"] == sorted_d[-1][0].numpy(), axis=1)
    e = np.abs(out["rgb"] - rgb[0].numpy()).max(-1)
    print(f"    oracle with the recorded numbers: {int(same.sum())}/{R} rays with bit-identical depths, max |rgb| error {e.max():.2e}; "
          f"change against the deterministic sampler: {np.abs(rgb[0].numpy() - rf['rgb']).max():.2e}")
    _check(f"{tag}.rgb", out["rgb"], rgb[0].numpy(), 1e-4)
    assert same.sum() >= 0.3 * R   # (the rest differ in the last bit of a few depths: the oracle's cdf is a float64 running sum rounded per element)
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), V=np.int64(V), u=np.stack(drawn), d_all=sorted_d[-1][0].numpy(), rgb=rgb[0].numpy(),
                        depth_volume=depth[0].numpy(), mask_volume=ex["mask_volume"][0].numpy(), normals_volume=ex["normals_volume"][0].numpy())


def gen_grad_fixture(tag, render_tag, V, mlp_state):
    """Training-side parity: the reference renderer WITH autograd on the rays of an
    existing render fixture, a fixed scalar loss, and d loss / d parameter for every model parameter."""
    import torch
    print(f"[{tag}] gradients of the reference renderer, rays of {render_tag}")
    rf = np.load(os.path.join(GOLDEN, f"{render_tag}.npz"))
    mesh = synthetic.fibonacci_blob(V)
    model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state)
    rays_o, rays_d = rf["rays_o"], rf["rays_d"]
    kw = dict(kw)
    kw.update(rayschunk=rays_o.shape[0], white_bkgd=bool(rf["white_bkgd"]), N_samples=int(rf["N_samples"]),
              N_importance=int(rf["N_samples"]), perturb=False, calc_normal=True)
    rng = np.random.default_rng(77)
    w_rgb = rng.uniform(0.5, 1.5, (rays_o.shape[0], 3)).astype(np.float32)
    w_n = rng.uniform(-1.0, 1.0, (rays_o.shape[0], 3)).astype(np.float32)
    model.train()
    rgb, depth, ex = renderer(torch.from_numpy(rays_o)[None], torch.from_numpy(rays_d)[None], detailed_output=False, **kw)
    loss = (rgb[0] * torch.from_numpy(w_rgb)).sum() + 0.1 * depth.sum() + 0.05 * ex["mask_volume"].sum() \
        + 0.02 * (ex["normals_volume"][0] * torch.from_numpy(w_n)).sum()
    loss.backward()
    grads = {}   # large matrices: Frobenius norm + the 48 rows of largest gradient norm (keeps the fixture small)
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().numpy().astype(np.float32)
        grads["norm." + name] = np.float32(np.linalg.norm(g.astype(np.float64)))
        if g.ndim == 2 and g.size > 4096:
            rows = np.sort(np.argsort(-np.linalg.norm(g, axis=1))[:48]).astype(np.int32)
            grads["rows." + name] = rows
            g = g[rows]
        grads["grad." + name] = g
    print(f"    loss {float(loss):.6f}; {len(grads)} parameter gradients; |grad ln_s| = {abs(float(grads['grad.ln_s'])):.4e}")
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), loss=np.float32(loss.item()), w_rgb=w_rgb, w_n=w_n,
                        rgb=rgb[0].detach().numpy(), **grads)


def gen_scale_fixture(tag="render_v140k_dtu", V=140_000, n_rays=1536, H=800, W=800, mlp_state=None, s_value=200.0,
                      n_samples=64, n_importance=64, white_bkgd=False, ckpt=None):
    """Headline-scale pin (BASELINE configs[1] shape, SURVEY 8d scene S-DTU): `n_rays` strided rays of
    frame 0 of the 800x800 orbit rendered by the IMPORTED REFERENCE at V = 140 000, with the stages a
    diverging ray can be traced through (near/far, coarse SDF, sorted depths after every up-sampling
    iteration, final SDF) and the reference's own sensitivity to a 1-ulp nudge of the ray directions.

    K-NN stand-in for FRNN: kd-tree candidates re-ranked under the declared fp32 arithmetic
    (oracle/knn.py:knn_kdtree), checked here against the brute-force declaration on a sample of
    the very query points the render issued."""
    import time
    import torch
    from scipy.spatial import cKDTree
    from oracle import knn as oknn
    print(f"[{tag}] V={V} rays={n_rays} of {H}x{W}")
    mesh = synthetic.fibonacci_blob(V)
    model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state, s_value=s_value, ckpt=ckpt)
    if ckpt is not None:                     # (the digest the test checks its own copy of the checkpoint against)
        mlp_state = ckpt_mlp_state(ckpt)
    import frnn as frnn_stub                 # oracle/refimport/stubs/frnn.py
    import models.renderer as ref_renderer   # reference
    tree = cKDTree(mesh.vertices.astype(np.float64))
    queries = []

    def knn_fn(q, v, K):
        queries.append(q)
        return oknn.knn_kdtree(q, v, K, tree=tree)

    old_knn = frnn_stub.KNN_FN[0]
    frnn_stub.KNN_FN[0] = knn_fn
    o_all, d_all_ = synthetic.camera_rays(synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W), H, W)
    sel = np.linspace(0, H * W - 1, n_rays).astype(np.int64)
    rays_o, rays_d = o_all[sel], d_all_[sel]
    kw = dict(kw)
    kw.update(rayschunk=n_rays, calc_normal=True, N_samples=n_samples, N_importance=n_importance, perturb=False, white_bkgd=white_bkgd)

    def run(rd, trace):
        """one reference render; trace (dict or None) receives the per-stage arrays"""
        rec = {"sdf_calls": [], "sorted": []}
        orig_bnf, orig_fdo, orig_fwn, orig_sort = (ref_renderer.compute_bounded_near_far, model.forward_density_only,
                                                   model.forward_with_nablas, torch.sort)

        def spy_bnf(*a, **k):
            near, far = orig_bnf(*a, **k)
            rec["near"], rec["far"] = near.detach().clone(), far.detach().clone()
            return near, far

        def spy_fdo(xyz):
            out = orig_fdo(xyz)
            rec["sdf_calls"].append(out.detach().clone())
            return out

        def spy_fwn(xyz):
            rec["pts"] = xyz.detach().clone()
            return orig_fwn(xyz)

        def spy_sort(x, *a, **k):
            out = orig_sort(x, *a, **k)
            rec["sorted"].append(out[0].detach().clone())
            return out

        ref_renderer.compute_bounded_near_far, model.forward_density_only, model.forward_with_nablas = spy_bnf, spy_fdo, spy_fwn
        torch.sort = spy_sort
        try:
            with torch.no_grad():
                rgb, depth, ex = renderer(torch.from_numpy(rays_o)[None], torch.from_numpy(rd)[None], detailed_output=True, **kw)
        finally:
            ref_renderer.compute_bounded_near_far, model.forward_density_only, model.forward_with_nablas = orig_bnf, orig_fdo, orig_fwn
            torch.sort = orig_sort
        out = {k: v[0].detach().numpy() for k, v in ex.items()}
        if trace is not None:
            assert len(rec["sdf_calls"]) == 5 and len(rec["sorted"]) == 4, (len(rec["sdf_calls"]), len(rec["sorted"]))
            trace["near_far"] = np.concatenate([rec["near"][0].numpy(), rec["far"][0].numpy()], axis=-1).astype(np.float32)
            trace["sdf_coarse"] = rec["sdf_calls"][0][0, ..., 0].numpy().reshape(n_rays, -1).astype(np.float32)
            for i, d in enumerate(rec["sorted"]):
                trace[f"d_iter{i + 1}"] = d[0].numpy().astype(np.float32)
            trace["pts"] = rec["pts"].numpy().reshape(n_rays, -1, 3)
        return out

    trace = {}
    t0 = time.perf_counter()
    ref = run(rays_d, trace)
    t_ref = time.perf_counter() - t0
    print(f"    reference render: {t_ref:.1f} s ({n_rays / t_ref:.0f} rays/s on {os.cpu_count()} cores, torch threads {torch.get_num_threads()})")
    d_all = trace["d_iter4"]
    assert d_all.shape == (n_rays, n_samples + n_importance)
    acc = ref["mask_volume"]
    stats = {"rays_acc_eq_0": int((acc == 0).sum()), "rays_acc_lt_1e-3": int((acc < 1e-3).sum()),
             "rays_partial(1e-3..0.999)": int(((acc >= 1e-3) & (acc <= 0.999)).sum()), "rays_acc_gt_0.999": int((acc > 0.999).sum()),
             "rgb_std": float(ref["rgb"].std()), "rgb_std_opaque_rays": float(ref["rgb"][acc > 0.999].std()) if (acc > 0.999).any() else None,
             "zero_weight_midpoint_fraction": float((ref["visibility_weights"] == 0).mean())}
    print("    scene:", stats)
    REPORT[f"{tag}.scene"] = stats
    # the kd-tree stand-in == the brute-force declaration on (a sample of) the queries this render issued
    qs = np.concatenate([q.reshape(-1, 3) for q in queries])
    pick = np.random.default_rng(5).choice(qs.shape[0], 40000, replace=False)
    bi, bd = oknn.knn_bruteforce(qs[pick], mesh.vertices, 8)
    ki, kd = oknn.knn_kdtree(qs[pick], mesh.vertices, 8, tree=tree)
    assert np.array_equal(bi, ki) and np.array_equal(bd, kd), "kd-tree + re-rank differs from the declared brute force"
    print(f"    K-NN stand-in == brute-force declaration on 40000 of the {qs.shape[0]} query points of this render")
    # the reference's own conditioning: the same render with every ray direction moved by 1 ulp
    queries.clear()
    ref2 = run(np.nextafter(rays_d, np.float32(10), dtype=np.float32), None)
    frnn_stub.KNN_FN[0] = old_knn
    self_err = np.abs(ref2["rgb"] - ref["rgb"]).max(-1).astype(np.float32)
    print(f"    reference vs itself (directions + 1 ulp): {int((self_err > 1e-4).sum())}/{n_rays} rays > 1e-4, "
          f"max {self_err.max():.2e}, median {np.median(self_err):.1e}, PSNR {compare.psnr(ref2['rgb'], ref['rgb']):.1f} dB")
    # oracle restatement on the same rays (report; the V=3000 fixtures gate it tightly)
    orc = oracle_from_reference(model, mesh)
    orc.knn_fn = lambda q, v, K: oknn.knn_kdtree(q, v, K, tree=tree)
    out = orender.render_rays(orc, rays_o, rays_d, orender.RenderConfig(calc_normal=True, N_samples=n_samples, N_importance=n_importance, white_bkgd=white_bkgd), detailed=True)
    e = np.abs(out["rgb"] - ref["rgb"]).max(-1)
    same = np.all(out["d_all"] == d_all, axis=1)
    print(f"    oracle vs reference: {int((e > 1e-4).sum())}/{n_rays} rays > 1e-4, median {np.median(e):.1e}, PSNR {compare.psnr(out['rgb'], ref['rgb']):.1f} dB; "
          f"rays with bit-identical sample sets {int(same.sum())}, max err among them {float(e[same].max()) if same.any() else float('nan'):.2e}")
    REPORT[f"{tag}.oracle_vs_reference"] = {"rays_gt_1e-4": int((e > 1e-4).sum()), "median": float(np.median(e)),
                                            "psnr_db": float(compare.psnr(out["rgb"], ref["rgb"])),
                                            "identical_sample_sets": int(same.sum()),
                                            "max_err_identical": float(e[same].max()) if same.any() else None}
    REPORT[f"{tag}.reference_self_sensitivity_1ulp"] = {"rays_gt_1e-4": int((self_err > 1e-4).sum()), "max": float(self_err.max()),
                                                       "psnr_db": float(compare.psnr(ref2["rgb"], ref["rgb"]))}
    assert same.any() and float(e[same].max()) <= 1e-5 and (e > 1e-4).mean() <= (self_err > 1e-4).mean() + 0.01
    np.savez_compressed(
        os.path.join(GOLDEN, f"{tag}.npz"),
        V=np.int64(V), H=np.int64(H), W=np.int64(W), frame=np.int64(0), sel=sel,
        rays_o=rays_o, rays_d=rays_d, near_far=trace["near_far"], sdf_coarse=trace["sdf_coarse"],
        d_iter1=trace["d_iter1"], d_iter2=trace["d_iter2"], d_iter3=trace["d_iter3"], d_all=d_all,
        sdf_all=ref["implicit_surface"].astype(np.float32),
        rgb=ref["rgb"], depth_volume=ref["depth_volume"], mask_volume=ref["mask_volume"], normals_volume=ref["normals_volume"],
        self_err_1ulp=self_err, s=np.float32(model.forward_s().item()), state_sha256=np.array(state_digest(mlp_state) if mlp_state is not None else ""),
        N_samples=np.int64(n_samples), N_importance=np.int64(n_importance), white_bkgd=np.int64(white_bkgd),
    )


def ckpt_mlp_state(path):
    """EVERY tensor of a utils/checkpoints.py file's "model" entry (MLPs, code tables, indicator vectors, ln_s) as numpy: what state_digest
    hashes for a checkpoint-backed fixture (the test loads its own copy of the file and must arrive at the same digest)."""
    import torch
    sd = torch.load(path, map_location="cpu")["model"]
    return {k: v.numpy() for k, v in sd.items()}


TRAINED_CKPT = os.path.join(GOLDEN, "trained_v140k.pt")


def gen_surf_sensitivity(tag="render_v140k_surf", mlp_state=None, s_value=400.0, n_seeds=8, n_time=3, timing=True, ckpt=None):
    """The reference's OWN spread on the headline fixture, and its own speed.

    (a) `n_seeds` independent last-bit perturbations of the fixture's ray directions -- seed 0: every component one ulp up (the run
        the fixture itself carries as self_err_1ulp), seeds 1..: every component moved by -1 / 0 / +1 ulp at random -- each rendered by
        the imported reference exactly as the fixture was; per seed the number of rays whose rgb moves by more than 1e-4 and the
        largest move.  The product's end-to-end gate is `count <= max over seeds`, `max error <= 2 x max over seeds`
        (tests/test_gpu_parity.py).  Written to tests/golden/<tag>_sens.npz; the pinned <tag>.npz is only READ (and checked: the
        unperturbed render must reproduce its rgb bit for bit).
    (b) the un-spied production call renderer(rays, detailed_output=False, **render_kwargs_test) on the first 512 fixture rays,
        `n_time` repeats, median: REPORT.json["reference_timing"] -- what bench.py quotes as the reference's CPU rays/s."""
    import time
    import torch
    from scipy.spatial import cKDTree
    from oracle import knn as oknn
    load_before = os.getloadavg()      # before this process has done any work of its own
    if timing and not n_seeds and load_before[0] > 1.0:   # the timing-only mode writes a RECORD: refuse on a busy machine
        raise SystemExit(f"reftime: load average {load_before[0]:.2f} > 1 -- run it on an otherwise idle machine")
    f = np.load(os.path.join(GOLDEN, f"{tag}.npz"))
    V, n_rays = int(f["V"]), f["rays_o"].shape[0]
    print(f"[{tag}_sens] V={V} rays={n_rays}, {n_seeds} perturbation seeds")
    mesh = synthetic.fibonacci_blob(V)
    model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state, s_value=s_value, ckpt=ckpt)
    if ckpt is not None:
        mlp_state = ckpt_mlp_state(ckpt)
    import frnn as frnn_stub
    tree = cKDTree(mesh.vertices.astype(np.float64))
    old_knn = frnn_stub.KNN_FN[0]
    frnn_stub.KNN_FN[0] = lambda q, v, K: oknn.knn_kdtree(q, v, K, tree=tree)
    kw = dict(kw)
    n_s = int(f["N_samples"]) if "N_samples" in f.files else 64          # (fixtures written before these keys existed: the headline shape)
    n_i = int(f["N_importance"]) if "N_importance" in f.files else 64
    kw.update(rayschunk=n_rays, calc_normal=True, N_samples=n_s, N_importance=n_i, perturb=False,
              white_bkgd=bool(f["white_bkgd"]) if "white_bkgd" in f.files else False)
    rays_o, rays_d = f["rays_o"], f["rays_d"]

    aux_keys = ("depth_volume", "mask_volume", "normals_volume")

    def render(rd, n=None, aux=None):
        with torch.no_grad():
            rgb, _, ex = renderer(torch.from_numpy(rays_o[:n])[None], torch.from_numpy(rd[:n])[None], detailed_output=False, **dict(kw, rayschunk=n or n_rays))
        if aux is not None:
            aux.update({k: ex[k][0].numpy() for k in aux_keys})
        return rgb[0].numpy()

    try:
        base_aux = {}
        base = render(rays_d, aux=base_aux)
        assert np.array_equal(base, f["rgb"]), "the unperturbed reference render does not reproduce the pinned fixture"
        assert all(np.array_equal(base_aux[k], f[k]) for k in aux_keys)
        up, down = np.nextafter(rays_d, np.float32(10), dtype=np.float32), np.nextafter(rays_d, np.float32(-10), dtype=np.float32)
        errs, counts, maxes = [], [], []
        aux_errs = {k: [] for k in aux_keys}     # the same for depth / acc / normals (round 6: their gates are paired too)
        for seed in range(n_seeds):
            if seed == 0:
                rd = up
            else:
                pick = np.random.default_rng(1000 + seed).integers(-1, 2, rays_d.shape)
                rd = np.where(pick > 0, up, np.where(pick < 0, down, rays_d)).astype(np.float32)
            aux = {}
            e = np.abs(render(rd, aux=aux) - base).max(-1).astype(np.float32)
            for k in aux_keys:
                aux_errs[k].append(np.abs(aux[k] - base_aux[k]).reshape(n_rays, -1).max(-1).astype(np.float32))
            errs.append(e)
            counts.append(int((e > 1e-4).sum()))
            maxes.append(float(e.max()))
            print(f"    seed {seed}: {counts[-1]}/{n_rays} rays > 1e-4, max {maxes[-1]:.2e}, median {np.median(e):.1e}")
        assert not n_seeds or np.array_equal(errs[0], f["self_err_1ulp"]), "seed 0 is the fixture's own 1-ulp run"
        if timing:
            step = max(1, n_rays // 512)                     # 512 rays strided over the whole frame, like bench.py's CPU baselines
            to, td = np.ascontiguousarray(rays_o[::step][:512]), np.ascontiguousarray(rays_d[::step][:512])
            n_t = to.shape[0]

            def timed_call(n):
                with torch.no_grad():
                    renderer(torch.from_numpy(to[:n])[None], torch.from_numpy(td[:n])[None], detailed_output=False, **dict(kw, rayschunk=n))

            timed_call(8)
            times = []
            for _ in range(n_time):
                t0 = time.perf_counter()
                timed_call(n_t)
                times.append(time.perf_counter() - t0)
    finally:
        frnn_stub.KNN_FN[0] = old_knn
    if timing:
        med = float(np.median(times))
        REPORT["reference_timing"] = {"rays_per_s": n_t / med, "rays": n_t, "repeats": n_time, "seconds": [float(t) for t in times], "cores": os.cpu_count(),
                                      "torch_threads": int(torch.get_num_threads()), "V": V, "samples_per_ray": n_s + n_i,
                                      "loadavg_before_start": [float(x) for x in load_before], "loadavg_after": [float(x) for x in os.getloadavg()],
                                      "scene": tag, "rays_are": f"every {step}th of the fixture's {n_rays} rays (strided over frame 0 of the 800x800 orbit)",
                                      "call": "models/renderer.py SingleRenderer.forward(rays, detailed_output=False, **render_kwargs_test), unmodified; "
                                      "FRNN stand-in: scipy cKDTree candidates + declared fp32 re-rank (oracle/knn.py)"}
        print(f"    reference timing: {n_t} rays, median of {n_time}: {med:.2f} s = {n_t / med:.1f} rays/s ({os.cpu_count()} cores, {torch.get_num_threads()} torch threads)")
    if not n_seeds:
        return
    REPORT[f"{tag}.reference_self_sensitivity_seeds"] = {"rays_gt_1e-4": counts, "max": maxes, "n_rays": n_rays}
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}_sens.npz"), self_err=np.stack(errs), rays_gt_1e_4=np.asarray(counts, np.int64),
                        **{"self_err_" + k: np.stack(v) for k, v in aux_errs.items()},
                        max_err=np.asarray(maxes, np.float32), state_sha256=np.array(state_digest(mlp_state) if mlp_state is not None else ""))


def gen_painting_step_fixture(tag="painting_step_v3000", V=3000, mlp_state=None, n_paint=40, n_bg=56):
    """The texture-painting fine-tune step through the REFERENCE's Trainer.forward_painting (models/trainer.py:119-172): painted rays
    rendered with random colour directions (renderer.py:279-289), background rays with per-sample outputs for the distillation terms
    (stub teacher), compute_loss on their concatenation, backward.  torch.rand_like -- the random directions -- is replaced by a
    host-generator draw of the same shape (seed below) so that the GPU test can feed the product the very same numbers."""
    import torch
    print(f"[{tag}] reference Trainer.forward_painting + backward, V={V}")
    mesh = synthetic.fibonacci_blob(V)
    lw = {"img": 1.0, "mask": 0.1, "eikonal": 0.1, "distill_density": 1.0, "distill_color": 1.0, "indicator_reg": 0.001}
    model, kw_test, renderer, args = harness.build_reference(mesh, seed=0, mlp_state=mlp_state, overrides={"training:loss_weights": dict(lw)})
    from models.trainer import Trainer  # reference
    trainer = Trainer(model, loss_weights=dict(lw), teacher_model=None, device_ids=["cpu"])
    trainer.teacher_model = StubTeacher()
    H = W = 40
    o_all, d_all = synthetic.camera_rays(synthetic.orbit_pose(4), synthetic.pinhole_intrinsics(H, W, 1.0), H, W)
    rng = np.random.default_rng(41)
    pick = rng.permutation(H * W)[:n_paint + n_bg]
    ip, ib = pick[:n_paint], pick[n_paint:]
    arr = {"rays_o_paint": o_all[ip], "rays_d_paint": d_all[ip], "rays_o_bg": o_all[ib], "rays_d_bg": d_all[ib],
           "mask_paint": rng.uniform(0, 1, n_paint) > 0.2, "mask_bg": rng.uniform(0, 1, n_bg) > 0.4,
           "rgb_paint": rng.uniform(0, 1, (n_paint, 3)).astype(np.float32), "rgb_bg": rng.uniform(0, 1, (n_bg, 3)).astype(np.float32)}
    model_input = {k: torch.from_numpy(v) for k, v in arr.items() if not k.startswith("rgb_")}
    ground_truth = {k: torch.from_numpy(v) for k, v in arr.items() if k.startswith("rgb_")}
    kw = dict(kw_test)
    kw.pop("rayschunk", None)
    kw.update(perturb=False, calc_normal=True, N_samples=64, N_importance=64, rayschunk=4096)
    model.train()
    gen = torch.Generator().manual_seed(77)
    orig = torch.rand_like
    torch.rand_like = lambda t, **k: torch.rand(t.shape, generator=gen, dtype=t.dtype)
    try:
        ret = trainer.forward_painting(args, None, model_input, ground_truth, kw, 0, device="cpu")
    finally:
        torch.rand_like = orig
    losses = ret["losses"]
    losses["total"].backward()
    out = {"loss." + k: np.float32(v.item()) for k, v in losses.items()}
    for name, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().numpy().astype(np.float32)
        out["norm." + name] = np.float32(np.linalg.norm(g.astype(np.float64)))
        if g.ndim == 2 and g.size > 4096:
            rows = np.sort(np.argsort(-np.linalg.norm(g, axis=1))[:48]).astype(np.int32)
            out["rows." + name] = rows
            g = g[rows]
        out["grad." + name] = g
    print("    losses:", {k: round(float(v), 6) for k, v in out.items() if k.startswith("loss.")})
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), V=np.int64(V), rand_seed=np.int64(77), psnr=np.float32(ret["extras"]["psnr"].item()),
                        loss_weight_keys=np.array(sorted(lw)), loss_weight_vals=np.array([lw[k] for k in sorted(lw)], np.float32), **arr, **out)
    REPORT[f"{tag}.losses"] = {k: float(v) for k, v in out.items() if k.startswith("loss.")}


def gen_render_py_trace(tag="render_py_trace", H=60, W=80, n_views=3, V=3000):
    """What the reference's own driver does around the renderer (VERDICT r2 missing #7): `render.render_function`
    (/root/reference/render.py:99-260) is RUN here, unmodified, on a 6-pose data set object, with a recording stand-in for the
    renderer -- so the fixture holds exactly what render.py hands to `render_fn` (rays from rend_util.get_rays for its own spiral
    camera path, the keyword arguments: build_framework's render_kwargs_test + what render_function adds) and what it writes from
    the stand-in's return values (cv2.imwrite / imageio.imwrite / imageio.mimwrite payloads).  The GPU test replays these calls on
    the product renderer and reproduces the written images with the product's frame assembly."""
    import json
    import tempfile
    import types
    import torch
    harness._activate()
    import cv2 as cv2_stub
    import imageio as imageio_stub
    import render as ref_render                    # /root/reference/render.py
    from utils import rend_util as ref_rend_util   # reference
    mesh = synthetic.fibonacci_blob(V)
    _model, kw_test, _renderer, args = harness.build_reference(mesh, seed=0)
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = np.asarray(synthetic.pinhole_intrinsics(H, W), np.float32)[:3, :3]

    class FakeDataset:                                   # the attributes render_function reads (render.py:108-131)
        def __init__(self):
            self.H, self.W = H, W
            self.c2w_all = [torch.from_numpy(np.asarray(synthetic.orbit_pose(7 * i), np.float32)) for i in range(6)]

        def __getitem__(self, i):
            return i, {"intrinsics": torch.from_numpy(K.copy()), "c2w": self.c2w_all[i], "object_mask": torch.ones(H * W, dtype=torch.bool)}, \
                {"rgb": torch.zeros(H * W, 3)}

    calls, rays_calls, written = [], [], []
    rng = np.random.default_rng(17)

    def recording_render_fn(rays_o, rays_d, **kw):
        n = rays_o.shape[-2]
        rgb = torch.from_numpy(rng.random((1, n, 3), dtype=np.float32))
        depth = torch.from_numpy((rng.random((1, n), dtype=np.float32) * 3 + 0.5).astype(np.float32))
        normals = torch.from_numpy((rng.random((1, n, 3), dtype=np.float32) * 2 - 1).astype(np.float32))
        calls.append({"rays_o": rays_o.numpy().copy(), "rays_d": rays_d.numpy().copy(), "kw": dict(kw),
                      "rgb": rgb.numpy().copy(), "depth": depth.numpy().copy(), "normals": normals.numpy().copy()})
        return rgb, depth, {"normals_volume": normals, "mask_volume": torch.ones(1, n), "depth_volume": depth}

    orig_get_rays = ref_rend_util.get_rays

    def spy_get_rays(c2w, intrinsics, H_, W_, N_rays=-1):
        rays_calls.append({"c2w": c2w.numpy().copy(), "intrinsics": intrinsics.numpy().copy(), "H": H_, "W": W_, "N_rays": N_rays})
        return orig_get_rays(c2w, intrinsics, H_, W_, N_rays=N_rays)

    rargs = types.SimpleNamespace(dataset_split=None, background=None, downscale=1, H=None, H_scale=None, W=None, W_scale=None,
                                  camera_path="spiral", test_frame=None, spiral_rad=[], num_views=n_views, rayschunk=4096, outbase="trace",
                                  expname="trace", outdirectory=None, disable_rgb=False, fps=30, data=args.data)
    saved = (ref_render.get_data, ref_rend_util.get_rays, torch.Tensor.cuda, cv2_stub.imwrite, imageio_stub.imwrite, imageio_stub.mimwrite)
    cwd = os.getcwd()
    try:
        ref_render.get_data = lambda a, downscale=1: FakeDataset()
        ref_rend_util.get_rays = spy_get_rays
        torch.Tensor.cuda = lambda self, *a, **k: self
        cv2_stub.imwrite = lambda path, img: written.append(("cv2.imwrite", os.path.basename(path), np.asarray(img).copy()))
        imageio_stub.imwrite = lambda path, img: written.append(("imageio.imwrite", os.path.basename(path), np.asarray(img).copy()))
        imageio_stub.mimwrite = lambda path, imgs, **k: written.append(("imageio.mimwrite", os.path.basename(path), np.stack([np.asarray(i) for i in imgs])))
        os.chdir(tempfile.mkdtemp())
        ref_render.render_function(rargs, dict(kw_test), recording_render_fn)
    finally:
        os.chdir(cwd)
        (ref_render.get_data, ref_rend_util.get_rays, torch.Tensor.cuda, cv2_stub.imwrite, imageio_stub.imwrite, imageio_stub.mimwrite) = saved
    assert len(calls) == n_views == len(rays_calls)
    kw0 = calls[0]["kw"]
    assert all(c["kw"] == kw0 for c in calls)
    out = {"H": np.int64(H), "W": np.int64(W), "V": np.int64(V), "n_views": np.int64(n_views),
           "kwargs_json": np.array(json.dumps({k: (v if not isinstance(v, (np.floating, np.integer)) else v.item()) for k, v in kw0.items()}, sort_keys=True))}
    for i, (c, r) in enumerate(zip(calls, rays_calls)):
        out[f"c2w_{i}"], out[f"intrinsics_{i}"] = r["c2w"], r["intrinsics"]
        assert r["H"] == H and r["W"] == W and r["N_rays"] == -1
        for k in ("rays_o", "rays_d", "rgb", "depth", "normals"):
            out[f"{k}_{i}"] = c[k]
    names = []
    for j, (fn, name, arr) in enumerate(written):
        names.append(f"{fn}:{name}")
        out[f"written_{j}"] = arr
    out["written_names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), **out)
    print(f"[{tag}] {n_views} renderer calls recorded from render.render_function; kwargs {sorted(kw0)}; files written: {names}")
    REPORT[f"{tag}"] = {"renderer_kwargs": sorted(kw0), "written": names}


def state_digest(state) -> str:
    """sha256 over the MLP tensors of a state dict (sorted keys, raw fp32 bytes): the product side
    re-derives the surface scene's weights and must arrive at these very bytes."""
    import hashlib
    h = hashlib.sha256()
    for k in sorted(state):
        h.update(k.encode())
        h.update(np.ascontiguousarray(state[k], dtype=np.float32).tobytes())
    return h.hexdigest()


class StubTeacher:
    """Deterministic stand-in for the NeuS teacher of the distillation losses (models/trainer.py:211-221):
    teacher(xyz, dirs) -> (sdf [...], radiance [..., 3]).  Analytic, so the product-side test can rebuild it."""

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def __call__(self, xyz, dirs):
        import torch
        return torch.linalg.norm(xyz, dim=-1) - 0.75, torch.sigmoid(2.0 * dirs + xyz)


def gen_train_step_fixture(tag="train_step_v3000", V=3000, mlp_state=None, s_value=200.0, n_rays=96, HW=40, kdtree=False):
    """One optimisation step's forward + backward through the REFERENCE's Trainer (models/trainer.py:50-117,174-285):
    random pixel selection, render with autograd (calc_normal on: eikonal weight > 0), per-sample outputs for the
    distillation terms (stub teacher), every loss term, and d total / d parameter.  perturb is switched off
    (its torch.rand stream is device-specific); the pixel selection uses the CPU generator on both sides."""
    import torch
    print(f"[{tag}] reference Trainer.forward + backward, V={V}")
    mesh = synthetic.fibonacci_blob(V)
    lw = {"img": 1.0, "mask": 0.1, "eikonal": 0.1, "distill_density": 1.0, "distill_color": 1.0, "indicator_reg": 0.001}
    model, kw_test, renderer, args = harness.build_reference(mesh, seed=0, mlp_state=mlp_state, s_value=s_value,
                                                             overrides={"training:loss_weights": dict(lw), "data:N_rays": n_rays})
    from models.trainer import Trainer  # reference
    trainer = Trainer(model, loss_weights=dict(lw), teacher_model=None, device_ids=["cpu"])
    trainer.teacher_model = StubTeacher()
    H = W = HW
    c2w, K = synthetic.orbit_pose(9), synthetic.pinhole_intrinsics(H, W, 1.0)
    rng = np.random.default_rng(31)
    gt_rgb = rng.uniform(0, 1, (1, H * W, 3)).astype(np.float32)
    obj_mask = rng.uniform(0, 1, (1, H * W)) > 0.4
    kw = dict(kw_test)
    kw.pop("rayschunk", None)
    kw.update(perturb=False, calc_normal=True, H=H, W=W, N_samples=64, N_importance=64, rayschunk=4096)
    model_input = {"intrinsics": torch.from_numpy(K)[None], "c2w": torch.from_numpy(c2w)[None], "object_mask": torch.from_numpy(obj_mask)}
    ground_truth = {"rgb": torch.from_numpy(gt_rgb)}
    model.train()
    import contextlib

    def step(pose):
        """one Trainer.forward + backward; returns (ret, {loss.*, norm.*, rows.*, grad.*}, full gradients)"""
        for p_ in model.parameters():
            p_.grad = None
        torch.manual_seed(123)
        mi = dict(model_input, c2w=torch.from_numpy(pose)[None])
        with (_kdtree_knn(mesh) if kdtree else contextlib.nullcontext()):
            ret_ = trainer.forward(args, None, mi, ground_truth, kw, 0, device="cpu")
            ret_["losses"]["total"].backward()
        o = {"loss." + k: np.float32(v.item()) for k, v in ret_["losses"].items()}
        full = {}
        for name, p_ in model.named_parameters():
            if p_.grad is None:
                continue
            g = p_.grad.detach().numpy().astype(np.float32).copy()
            full[name] = g
            o["norm." + name] = np.float32(np.linalg.norm(g.astype(np.float64)))
            if g.ndim == 2 and g.size > 4096:
                rows = np.sort(np.argsort(-np.linalg.norm(g, axis=1))[:48]).astype(np.int32)
                o["rows." + name] = rows
                g = g[rows]
            o["grad." + name] = g
        return ret_, o, full

    ret, out, full = step(c2w)
    losses = ret["losses"]
    if kdtree:
        # the reference's own conditioning on a scene with a surface: the same step with the camera pose moved by 1 ulp (sample
        # placement at a sharp crossing is sensitive to the last bit, as for the render fixtures): loss and gradient deltas recorded
        _, out2, full2 = step(np.nextafter(c2w, np.float32(10), dtype=np.float32))
        for k in [k for k in out if k.startswith("loss.")]:
            out["self1ulp." + k] = np.float32(abs(float(out2[k]) - float(out[k])))
        for name in full:
            out["self1ulp.gradrel." + name] = np.float32(np.abs(full2[name] - full[name]).max() / max(np.abs(full[name]).max(), 1e-12))
            out["self1ulp.normrel." + name] = np.float32(abs(float(out2["norm." + name]) - float(out["norm." + name])) / max(float(out["norm." + name]), 1e-12))
        print("    reference vs itself (pose + 1 ulp): loss deltas", {k[14:]: float(v) for k, v in out.items() if k.startswith("self1ulp.loss.")})
        print("    reference vs itself (pose + 1 ulp): largest relative gradient deltas",
              sorted(((float(v), k[17:]) for k, v in out.items() if k.startswith("self1ulp.gradrel.")), reverse=True)[:5])
    print("    losses:", {k: round(float(v), 6) for k, v in out.items() if k.startswith("loss.")})
    print(f"    psnr {float(ret['extras']['psnr']):.3f}, 1/s {float(ret['extras']['scalars']['1/s']):.5f}, "
          f"|grad ln_s| {abs(float(out['grad.ln_s'])):.3e}")
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), V=np.int64(V), H=np.int64(H), W=np.int64(W), c2w=c2w, intrinsics=K,
                        gt_rgb=gt_rgb, object_mask=obj_mask, select_inds=ret["extras"]["select_inds"].numpy(),
                        psnr=np.float32(ret["extras"]["psnr"].item()), rgb=ret["extras"]["mask_volume_clipped"].detach().numpy(),
                        N_rays=np.int64(n_rays), s=np.float32(model.forward_s().item()),
                        loss_weight_keys=np.array(sorted(lw)), loss_weight_vals=np.array([lw[k] for k in sorted(lw)], np.float32), **out)
    if tag != "train_step_v3000":
        return
    # compute_loss alone on fixed tensors, the three masking variants (CPU-side test of the product's Trainer)
    rng = np.random.default_rng(32)
    B, R, N = 1, 50, 12
    rgb = torch.from_numpy(rng.uniform(0, 1, (B, R, 3)).astype(np.float32))
    tgt = torch.from_numpy(rng.uniform(0, 1, (B, R, 3)).astype(np.float32))
    acc = torch.from_numpy(rng.uniform(0, 1, (B, R)).astype(np.float32))
    nab = torch.from_numpy(rng.normal(0, 1, (B, R, N, 3)).astype(np.float32))
    xyz = torch.from_numpy(rng.uniform(-1, 1, (B, R, N - 1, 3)).astype(np.float32))
    dirs = torch.nn.functional.normalize(torch.from_numpy(rng.normal(0, 1, (B, R, N - 1, 3)).astype(np.float32)), dim=-1)
    dens = torch.from_numpy(rng.normal(0, 0.2, (B, R, N - 1, 1)).astype(np.float32))
    cols = torch.from_numpy(rng.uniform(0, 1, (B, R, N - 1, 3)).astype(np.float32))
    m = torch.from_numpy(rng.uniform(0, 1, (B, R)) > 0.3)
    mi = torch.from_numpy(rng.uniform(0, 1, (B, R)) > 0.2)
    cl = {"rgb": rgb.numpy(), "target": tgt.numpy(), "mask_volume": acc.numpy(), "implicit_nablas": nab.numpy(), "xyz": xyz.numpy(),
          "dirs": dirs.numpy(), "density": dens.numpy(), "colors": cols.numpy(), "mask": m.numpy(), "mask_ignore": mi.numpy(),
          "indicator_vector": model.indicator_vector.detach().numpy(), "vertex_normals": model.mesh_grid.get_vertex_normal_torch().numpy(),
          "s": np.float32(model.forward_s().item())}
    with torch.no_grad():
        for vname, mk, mik in (("both", m, mi), ("mask_only", m, None), ("ignore_only", None, mi), ("none", None, None)):
            ex = {"mask_volume": acc.clone(), "implicit_nablas": nab, "xyz": xyz, "dirs": dirs, "density": dens, "colors": cols}
            r = trainer.compute_loss(args, rgb, tgt, ex, mask=mk, mask_ignore=mik, use_eikonal_loss=True, use_distill_loss=True,
                                     use_indicator_reg=True)
            for k, v in r["losses"].items():
                cl[f"{vname}.{k}"] = np.float32(v.item())
            cl[f"{vname}.psnr"] = r["extras"]["psnr"].numpy().astype(np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "trainer_compute_loss.npz"), **cl)


TRAIN_LOOP_LR = {"default": 5.0e-4, "color_features": 2.0e-3, "views_linears": 1.0e-4}
TRAIN_LOOP_SCHED = {"type": "warmupcosine", "warmup_steps": 2}


def gen_train_loop_fixture(tag="train_loop_v3000", V=3000, mlp_state=None, n_iters=6, num_iters=8, n_rays=96, HW=40):
    """The reference's OWN optimisation loop body, several iterations in a row: train.train (train.py:165-195: Trainer.forward, mean of
    every loss, zero_grad, backward, optimizer.step, scheduler.step(it)) with the optimizer of models/base.py:578-616 (dict learning rates:
    a per-parameter group, a per-module group, the rest) and the scheduler of models/base.py:648-676 (warmupcosine as a LambdaLR, stepped
    with the iteration number), ln_s frozen as train.py:290 does.  Records every iteration's losses and learning rates, the parameter
    groups, and the parameters after the last iteration.  Only change: Trainer.forward's device argument (default "cuda") is "cpu"."""
    import torch
    print(f"[{tag}] reference train.train x {n_iters} (get_optimizer / get_scheduler of models/base.py), V={V}")
    mesh = synthetic.fibonacci_blob(V)
    lw = {"img": 1.0, "mask": 0.1, "eikonal": 0.1, "distill_density": 1.0, "distill_color": 1.0, "indicator_reg": 0.001}
    model, kw_test, renderer, args = harness.build_reference(
        mesh, seed=0, mlp_state=mlp_state,
        overrides={"training:loss_weights": dict(lw), "data:N_rays": n_rays, "training:lr": dict(TRAIN_LOOP_LR),
                   "training:scheduler": dict(TRAIN_LOOP_SCHED), "training:num_iters": num_iters})
    import train as ref_train                                  # reference train.py
    from models.base import get_optimizer, get_scheduler       # reference
    from models.trainer import Trainer                         # reference
    trainer = Trainer(model, loss_weights=dict(lw), teacher_model=None, device_ids=["cpu"])
    trainer.teacher_model = StubTeacher()

    class OnCpu:   # train.train calls trainer.forward without a device (default "cuda")
        def forward(self, *a, **k):
            return trainer.forward(*a, device="cpu", **k)

    H = W = HW
    K = synthetic.pinhole_intrinsics(H, W, 1.0)
    rng = np.random.default_rng(41)
    gt_rgb = rng.uniform(0, 1, (1, H * W, 3)).astype(np.float32)
    obj_mask = rng.uniform(0, 1, (1, H * W)) > 0.4
    poses = np.stack([synthetic.orbit_pose(3 + 2 * i) for i in range(n_iters)]).astype(np.float32)
    kw = dict(kw_test)
    kw.pop("rayschunk", None)
    kw.update(perturb=False, calc_normal=True, H=H, W=W, N_samples=64, N_importance=64, rayschunk=4096)
    ground_truth = {"rgb": torch.from_numpy(gt_rgb)}
    model.train()
    model.ln_s.requires_grad = args.training.setdefault("required_grad_lns", False)   # train.py:290
    names = {id(p_): n for n, p_ in model.named_parameters()}
    optimizer = get_optimizer(args, model)
    groups = [[names[id(p_)] for p_ in g["params"]] for g in optimizer.param_groups]
    scheduler = get_scheduler(args, optimizer, last_epoch=-1)
    start = {n: p_.detach().numpy().copy() for n, p_ in model.named_parameters()}
    out = {"group_lr0": np.array([g["initial_lr"] for g in optimizer.param_groups], np.float64)}
    for gi, g in enumerate(groups):
        out[f"group{gi}.names"] = np.array(g)
    lr_used, lr_next, sel = [], [], []
    loss_keys = None
    loss_rows = []
    for it in range(n_iters):
        torch.manual_seed(500 + it)
        mi = {"intrinsics": torch.from_numpy(K)[None], "c2w": torch.from_numpy(poses[it])[None], "object_mask": torch.from_numpy(obj_mask)}
        lr_used.append([g["lr"] for g in optimizer.param_groups])
        losses, extras = ref_train.train(args, it, None, mi, ground_truth, kw, OnCpu(), optimizer, scheduler)
        lr_next.append([g["lr"] for g in optimizer.param_groups])
        if loss_keys is None:
            loss_keys = sorted(losses)
        loss_rows.append([float(losses[k].item()) for k in loss_keys])
        sel.append(extras["select_inds"].numpy())
        print(f"    it {it}: lr {lr_used[-1]} total {float(losses['total']):.6f} img {float(losses['loss_img']):.6f}")
    out.update(loss_keys=np.array(loss_keys), losses=np.array(loss_rows, np.float64), lr_used=np.array(lr_used, np.float64),
               lr_next=np.array(lr_next, np.float64), select_inds=np.stack(sel))
    for n, p_ in model.named_parameters():
        end = p_.detach().numpy()
        d = (end.astype(np.float64) - start[n].astype(np.float64))
        out["dnorm." + n] = np.float64(np.linalg.norm(d))
        out["dmax." + n] = np.float64(np.abs(d).max())
        if end.ndim == 2 and end.size > 4096:
            rows = np.sort(np.argsort(-np.linalg.norm(d, axis=1))[:48]).astype(np.int32)
            out["rows." + n] = rows
            out["end." + n] = end[rows].copy()
            out["start." + n] = start[n][rows].copy()
        else:
            out["end." + n] = end.copy()
            out["start." + n] = start[n].copy()
    print("    largest parameter moves:", sorted(((float(v), k[5:]) for k, v in out.items() if k.startswith("dmax.")), reverse=True)[:4])
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), V=np.int64(V), H=np.int64(H), W=np.int64(W), poses=poses, intrinsics=K,
                        gt_rgb=gt_rgb, object_mask=obj_mask, N_rays=np.int64(n_rays), n_iters=np.int64(n_iters), num_iters=np.int64(num_iters),
                        lr_keys=np.array(sorted(TRAIN_LOOP_LR)), lr_vals=np.array([TRAIN_LOOP_LR[k] for k in sorted(TRAIN_LOOP_LR)], np.float64),
                        warmup_steps=np.int64(TRAIN_LOOP_SCHED["warmup_steps"]),
                        loss_weight_keys=np.array(sorted(lw)), loss_weight_vals=np.array([lw[k] for k in sorted(lw)], np.float32), **out)


def gen_surface_fixture(tag="surface_v3000", V=3000, mlp_state=None):
    """models/ray_casting.py of the reference (dead code there: imported nowhere) run on the reference's NeuMesh
    field: root_finding_surface_points (256 proposals + 8 secant steps) and sphere_tracing_surface_points, on the
    rays of the render fixture.  The SDF handed in is model.forward_density_only."""
    import torch
    print(f"[{tag}] reference ray_casting.py on the reference field, V={V}")
    mesh = synthetic.fibonacci_blob(V)
    model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state)
    import models.ray_casting as rc  # reference
    rf = np.load(os.path.join(GOLDEN, "render_v3000_dtu.npz"))
    ro = torch.from_numpy(rf["rays_o"])[None]
    rd = torch.nn.functional.normalize(torch.from_numpy(rf["rays_d"]), dim=-1)[None]

    def sdf(p):
        with torch.no_grad():
            return model.forward_density_only(p).squeeze(-1)

    out = {}
    # (the default-init field is negative everywhere, ~-0.09..-0.05 along these rays: the level sets chosen through
    #  logit_tau are the ones the rays actually cross)
    for name, tau in (("tau_a", -0.085), ("tau_b", -0.075)):
        d, pt, m, msc = rc.root_finding_surface_points(sdf, ro.clone(), rd.clone(), near=0.8, far=3.6, batched=True, N_steps=256,
                                                       logit_tau=tau, method="secant", N_secant_steps=8, fill_inf=False)
        out.update({f"{name}.d": d[0].numpy(), f"{name}.pt": pt[0].numpy(), f"{name}.mask": m[0].numpy(), f"{name}.sign_change": msc[0].numpy(),
                    f"{name}.tau": np.float32(tau)})
        print(f"    root finding tau={tau}: {int(m.sum())}/{m.numel()} rays hit, {int(msc.sum())} with a sign change; depth of hits "
              f"{float(d[m].min()) if m.any() else float('nan'):.3f}..{float(d[m].max()) if m.any() else float('nan'):.3f}")

    class Surf:   # the same field shifted to the -0.08 level set, so that the marching has something to converge to
        def forward(self, p):
            return sdf(p) + 0.08

    d, pt, m = rc.sphere_tracing_surface_points(Surf(), ro.clone(), rd.clone(), near=0.8, far=3.6, batched=True, N_iters=20)
    out.update({"st.d": d[0].numpy(), "st.pt": pt[0].numpy(), "st.mask": m[0].numpy()})
    print(f"    sphere tracing: {int(m.sum())}/{m.numel()} rays still inside [0, far] after 20 steps")
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), V=np.int64(V), rays_o=rf["rays_o"], rays_d=rf["rays_d"], near=np.float32(0.8),
                        far=np.float32(3.6), **out)


def _kdtree_knn(mesh):
    """Context manager: FRNN stand-in = kd-tree candidates re-ranked under the declared fp32 arithmetic (oracle/knn.py:knn_kdtree; the
    headline-scale fixtures check it against the brute-force declaration).  Yields the list the issued query arrays are appended to."""
    import contextlib
    from scipy.spatial import cKDTree
    from oracle import knn as oknn
    import frnn as frnn_stub

    @contextlib.contextmanager
    def ctx():
        tree = cKDTree(mesh.vertices.astype(np.float64))
        queries = []

        def knn_fn(q, v, K):
            queries.append(q)
            return oknn.knn_kdtree(q, v, K, tree=tree)
        old = frnn_stub.KNN_FN[0]
        frnn_stub.KNN_FN[0] = knn_fn
        try:
            yield queries
        finally:
            frnn_stub.KNN_FN[0] = old
    return ctx()


def gen_texture_edit_fixture(tag="texture_edit_v3000", V=3000, mlp_state=None):
    """the reference's TextureEditableNeuMesh (editing/texture_neumesh/texture_neumesh.py:7-122) built on reference
    NeuMesh models -- one / two texture references, with and without rigid transforms (T_r_m_list) -- queried point-wise at the
    field fixture's points and rendered by the reference's SingleRenderer (as editing/texture_neumesh/texture_renderer.py:73-75 does)
    on the render fixture's rays.  The scene (masks, edited codes, transforms, reference colour networks) comes from
    neumesh_amd.synthetic.edit_scene / reference_color_state, which the product-side test calls with the same arguments."""
    import torch
    print(f"[{tag}] reference TextureEditableNeuMesh, V={V}")
    mesh = synthetic.fibonacci_blob(V)
    main, kw, _renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state)
    from editing.texture_neumesh.texture_neumesh import TextureEditableNeuMesh  # reference
    from models.renderer import SingleRenderer  # reference
    fx = np.load(os.path.join(GOLDEN, "field_v3000.npz"))
    rf = np.load(os.path.join(GOLDEN, "render_v3000_dtu.npz"))
    q, dirs = torch.from_numpy(fx["q"]), torch.from_numpy(fx["dirs"])
    ro, rd = torch.from_numpy(rf["rays_o"])[None], torch.from_numpy(rf["rays_d"])[None]
    kw = dict(kw)
    kw.update(rayschunk=ro.shape[1], calc_normal=True, N_samples=64, N_importance=64, perturb=False, white_bkgd=False)
    refs = []
    for i in range(2):
        m, *_ = harness.build_reference(mesh, seed=0, mlp_state=synthetic.reference_color_state(mlp_state, i))
        refs.append(m)
    with torch.no_grad():
        img0, dep0, _ = SingleRenderer(main)(ro, rd, detailed_output=False, **kw)
    out = {"V": np.int64(V), "main.rgb_render": img0[0].numpy(), "main.depth": dep0[0].numpy()}
    for name, n_ref, rotated in (("r1", 1, False), ("r2", 2, False), ("r2T", 2, True)):
        masks, feats, T_list = synthetic.edit_scene(mesh.vertices, n_ref, rotated)
        wrap = TextureEditableNeuMesh(main, refs[:n_ref], torch.from_numpy(masks), torch.from_numpy(feats),
                                      None if T_list is None else torch.FloatTensor(np.stack(T_list)))
        wrap.eval()
        sdf, rgb = wrap(q.clone(), dirs)
        with torch.no_grad():
            img, dep, ex = SingleRenderer(wrap)(ro, rd, detailed_output=False, **kw)
        painted = (torch.from_numpy(masks)[:, torch.from_numpy(fx["idx"].astype(np.int64))].any(-1)).any(0).numpy()   # any painted neighbour
        print(f"    {name}: points with a painted neighbour {painted.mean():.2f}; max |edited - main| point colour "
              f"{float((rgb.detach() - torch.from_numpy(fx['rgb'])).abs().max()):.3f}, render {float((img - img0).abs().max()):.4f}, acc mean {float(ex['mask_volume'].mean()):.3f}")
        assert torch.equal(dep, dep0) and float((img - img0).abs().max()) > 1e-3
        out.update({f"{name}.sdf": sdf.detach().numpy(), f"{name}.rgb": rgb.detach().numpy(), f"{name}.rgb_render": img[0].numpy(),
                    f"{name}.depth": dep[0].numpy(), f"{name}.acc": ex["mask_volume"][0].numpy(), f"{name}.normals": ex["normals_volume"][0].numpy(),
                    f"{name}.painted": painted, f"{name}.mask_sum": masks.sum(1).astype(np.int64),
                    f"{name}.feats_digest": np.array(state_digest({"f": feats}))})
    for i in range(2):
        out[f"ref{i}.state_sha256"] = np.array(state_digest(synthetic.reference_color_state(mlp_state, i)))
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), **out)


def gen_texture_edit_scale_fixture(tag="texture_edit_v140k_surf", V=140_000, n_rays=384, mlp_state=None, s_value=400.0):
    """Row f2 at the scale and on the scene the bench times it on: the reference's TextureEditableNeuMesh (two references with overlapping
    painted caps and rigid transforms) over reference NeuMesh models of the SURFACE scene at V = 140 000, rendered by the reference's
    SingleRenderer on the first `n_rays` rays of tests/golden/render_v140k_surf.npz (so that fixture's main-model pixels and 1-ulp
    self-sensitivity of the same rays are at hand)."""
    import torch
    print(f"[{tag}] reference TextureEditableNeuMesh on the surface scene, V={V}, rays={n_rays}")
    mesh = synthetic.fibonacci_blob(V)
    main, kw, _renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state, s_value=s_value)
    from editing.texture_neumesh.texture_neumesh import TextureEditableNeuMesh  # reference
    from models.renderer import SingleRenderer  # reference
    base = np.load(os.path.join(GOLDEN, "render_v140k_surf.npz"))
    rays_o, rays_d = base["rays_o"][:n_rays], base["rays_d"][:n_rays]
    ro, rd = torch.from_numpy(rays_o)[None], torch.from_numpy(rays_d)[None]
    kw = dict(kw)
    kw.update(rayschunk=n_rays, calc_normal=True, N_samples=64, N_importance=64, perturb=False, white_bkgd=False)
    refs = []
    for i in range(2):
        m, *_ = harness.build_reference(mesh, seed=0, mlp_state=synthetic.reference_color_state(mlp_state, i, gain=1.5), s_value=s_value)
        refs.append(m)
    masks, feats, T_list = synthetic.edit_scene(mesh.vertices, 2, True)
    wrap = TextureEditableNeuMesh(main, refs, torch.from_numpy(masks), torch.from_numpy(feats), torch.FloatTensor(np.stack(T_list)))
    wrap.eval()
    with _kdtree_knn(mesh), torch.no_grad():
        img, dep, ex = SingleRenderer(wrap)(ro, rd, detailed_output=False, **kw)
    d_main = np.abs(img[0].numpy() - base["rgb"][:n_rays]).max(-1)
    acc = ex["mask_volume"][0].numpy()
    print(f"    edited frame against the reference's main-model frame of the same rays: {int((d_main > 1e-2).sum())}/{n_rays} rays moved by > 1e-2 "
          f"(max {d_main.max():.3f}); depth identical: {bool(np.array_equal(dep[0].numpy(), base['depth_volume'][:n_rays]))}; opaque rays {int((acc > 0.999).sum())}")
    assert (d_main > 1e-2).sum() >= 0.1 * n_rays and np.array_equal(dep[0].numpy(), base["depth_volume"][:n_rays])
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), V=np.int64(V), n_rays=np.int64(n_rays), rgb=img[0].numpy(), depth_volume=dep[0].numpy(),
                        mask_volume=acc, normals_volume=ex["normals_volume"][0].numpy(), mask_sum=masks.sum(1).astype(np.int64),
                        state_sha256=np.array(state_digest(mlp_state)), s=np.float32(main.forward_s().item()))


def gen_deform_fixture(tag="deform_v3000", V=3000, mlp_state=None):
    """SURVEY 8f-2, geometry editing: the reference's deform_model (editing/render_geometry_editing.py:37-67, with kornia's
    angle_axis_to_rotation_matrix restated in oracle/refimport/stubs/kornia) run on a stretched + sheared mesh: the rotated
    indicator vectors it installs (incl. vertices whose normal flips exactly and vertices whose normal does not move), the field
    at the field fixture's points and a render of the render fixture's rays after the deformation; also with fix_indicator=True."""
    import torch
    harness._activate()
    import open3d as o3d_stub   # oracle/refimport/stubs/open3d
    print(f"[{tag}] reference deform_model, V={V}")
    base, dmesh, snap = synthetic.deformed_blob(synthetic.fibonacci_blob(V))
    fx = np.load(os.path.join(GOLDEN, "field_v3000.npz"))
    rf = np.load(os.path.join(GOLDEN, "render_v3000_dtu.npz"))
    q = torch.from_numpy(fx["q"])
    ro, rd = torch.from_numpy(rf["rays_o"])[None], torch.from_numpy(rf["rays_d"])[None]
    out = {"V": np.int64(V), "snap": snap.astype(np.int64), "base_normals": base.vertex_normals, "deformed_vertices": dmesh.vertices,
           "deformed_normals": dmesh.vertex_normals}
    for name, fix in (("rot", False), ("fix", True)):
        model, kw, renderer, _ = harness.build_reference(base, seed=0, mlp_state=mlp_state)
        from editing.render_geometry_editing import deform_model  # reference
        ind0 = model.indicator_vector.detach().clone()
        deform_model(o3d_stub.TriangleMesh(dmesh.vertices, dmesh.vertex_normals), model, "cpu", fix_indicator=fix)
        ind1 = model.indicator_vector.detach()
        kw = dict(kw)
        kw.update(rayschunk=ro.shape[1], calc_normal=True, N_samples=64, N_importance=64, perturb=False, white_bkgd=False)
        with torch.no_grad():
            ds, idx, w = model.compute_distance(q)
            sdf = model.forward_density_only(q)
            img, dep, ex = renderer(ro, rd, detailed_output=False, **kw)
        moved = (ind1 - ind0).norm(dim=-1)
        print(f"    {name}: indicator vectors moved by median {float(moved.median()):.3f}, max {float(moved.max()):.3f}; flipped "
              f"{(ind1[snap[:3]] + ind0[snap[:3]]).abs().max():.1e}, unmoved {(ind1[snap[3:]] - ind0[snap[3:]]).abs().max():.1e}; "
              f"acc mean {float(ex['mask_volume'].mean()):.3f}")
        if not fix:
            assert float((ind1[snap[:3]] + ind0[snap[:3]]).abs().max()) == 0.0 and float((ind1[snap[3:]] - ind0[snap[3:]]).abs().max()) == 0.0
            assert float(moved.median()) > 1e-3
        else:
            assert torch.equal(ind1, ind0)
        out.update({f"{name}.indicator": ind1.numpy(), f"{name}.ds": ds.numpy(), f"{name}.idx": idx.numpy().astype(np.int32), f"{name}.sdf": sdf.numpy(),
                    f"{name}.rgb_render": img[0].numpy(), f"{name}.depth": dep[0].numpy(), f"{name}.acc": ex["mask_volume"][0].numpy(),
                    f"{name}.normals": ex["normals_volume"][0].numpy()})
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), **out)


class _NeuSLike:
    """What the reference's surface_render expects of a model (models/ray_casting.py:268-288: `model.implicit_surface`,
    `model.forward(pts, view_dirs) -> (radiance, sdf, nablas)` -- the NeuS-style interface; the reference NeuMesh returns two values,
    which is why the module is dead code there), built from the reference NeuMesh's own methods."""

    def __init__(self, model):
        self.model = model

    def implicit_surface(self, pts):
        import torch
        with torch.no_grad():
            return self.model.forward_density_only(pts).squeeze(-1)

    def forward(self, pts, view_dirs):
        sdf, nab = self.model.forward_with_nablas(pts)
        _, rgb = self.model.forward(pts, view_dirs)
        return rgb.detach(), sdf.detach(), nab.detach()


def gen_surface_scale_fixture(tag="surface_v140k_surf", V=140_000, n_rays=512, H=800, W=800, mlp_state=None, s_value=400.0):
    """Headline-scale pin of row a16 / f4: the reference's models/ray_casting.py -- root_finding_surface_points (256 proposals +
    8 secant steps, level 0) and surface_render (colour / depth / nabla / normal at the first hit) -- on the reference NeuMesh field of
    the SURFACE scene at V = 140 000, `n_rays` strided rays of bench frame 0.  Per ray the fixture also holds the smallest |sdf| among
    the proposals up to (and including) the bracket of its first sign change: a ray can only change its mask / bracket under a field
    difference larger than that margin."""
    import torch
    print(f"[{tag}] reference ray_casting.py on the surface scene, V={V}, rays={n_rays}")
    mesh = synthetic.fibonacci_blob(V)
    model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp_state, s_value=s_value)
    import models.ray_casting as rc  # reference
    o_all, d_all_ = synthetic.camera_rays(synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W), H, W)
    sel = np.linspace(0, H * W - 1, n_rays).astype(np.int64)
    rays_o, rays_d = o_all[sel], d_all_[sel]
    ro = torch.from_numpy(rays_o)[None]
    rd = torch.nn.functional.normalize(torch.from_numpy(rays_d), dim=-1)[None]
    near, far, tau = 1.2, 3.2, 0.0
    adaptor = _NeuSLike(model)
    calls = []

    def sdf(p):
        v = adaptor.implicit_surface(p)
        calls.append(v.detach().clone())
        return v

    with _kdtree_knn(mesh):
        d, pt, m, msc = rc.root_finding_surface_points(sdf, ro.clone(), rd.clone(), near=near, far=far, batched=True, N_steps=256,
                                                       logit_tau=tau, method="secant", N_secant_steps=8, fill_inf=False)
        val = (calls[0] - tau)[0].numpy()                       # [R, 256]: the proposal values (first query of the routine)
        assert val.shape == (n_rays, 256)
        col, dep, ex = rc.surface_render(ro.clone(), torch.from_numpy(rays_d)[None], adaptor, calc_normal=True, batched=True,
                                         ray_casting_algo="root_finding",
                                         ray_casting_cfgs=dict(near=near, far=far, logit_tau=tau, fill_inf=False, N_steps=256, N_secant_steps=8))
        sdf_hit = np.abs(adaptor.implicit_surface(pt[m]).numpy())
        residual_all = adaptor.implicit_surface(pt)
    prod = val[:, :-1] * val[:, 1:]
    first = np.where((prod < 0).any(1), (prod < 0).argmax(1), 254)
    upto = np.arange(256)[None, :] <= (first[:, None] + 1)
    margin = np.where(upto, np.abs(val), np.inf).min(1).astype(np.float32)
    hit = m[0].numpy()
    print(f"    {int(hit.sum())}/{n_rays} rays hit, {int(msc.sum())} with a sign change; depth of hits {float(d[m].min()):.3f}..{float(d[m].max()):.3f}; "
          f"rays with margin < 1e-5: {int((margin < 1e-5).sum())}; |sdf| at the hits: median {float(np.median(sdf_hit)):.1e}, max {float(sdf_hit.max()):.1e}")
    assert torch.equal(ex["mask_surface"], m) and torch.equal(dep, d)
    np.savez_compressed(os.path.join(GOLDEN, f"{tag}.npz"), V=np.int64(V), H=np.int64(H), W=np.int64(W), sel=sel, rays_o=rays_o, rays_d=rays_d,
                        near=np.float32(near), far=np.float32(far), tau=np.float32(tau), d=d[0].numpy(), pt=pt[0].numpy(), mask=hit,
                        sign_change=msc[0].numpy(), margin=margin, first=first.astype(np.int32),
                        color=col[0].numpy(), nablas=ex["implicit_nablas"][0].numpy(), normals=ex["normals_surface"][0].numpy(),
                        residual=np.where(hit, np.abs(residual_all[0].numpy()), 0).astype(np.float32),   # |sdf| at the returned point
                        s=np.float32(model.forward_s().item()), state_sha256=np.array(state_digest(mlp_state)))


def gen_rays_fixture():
    """rend_util.get_rays of the reference (utils/rend_util.py:123-176) for a skewed pin-hole camera."""
    import torch
    harness._activate()
    from utils import rend_util  # reference
    H, W = 11, 17
    c2w = synthetic.orbit_pose(7)
    K = synthetic.pinhole_intrinsics(H, W, 0.9)
    K[0, 1] = 0.37          # skew, so every term of lift() is exercised
    K[1, 1] *= 1.1
    ro, rd, sel = rend_util.get_rays(torch.from_numpy(c2w)[None], torch.from_numpy(K)[None], H, W, N_rays=-1)
    o_o, o_d = orender.get_rays(c2w, K, H, W)
    _check("rays.rays_d", o_d, rd[0].numpy(), 3e-7)
    _check("rays.rays_o", o_o, ro[0].numpy(), 0.0)
    assert np.array_equal(sel[0].numpy(), np.arange(H * W))
    np.savez_compressed(os.path.join(GOLDEN, "rays_cam.npz"), c2w=c2w, intrinsics=K, H=np.int64(H), W=np.int64(W),
                        rays_o=ro[0].numpy(), rays_d=rd[0].numpy())


# perturbation seeds of the *_sens fixtures.  Round 6: 32 (was 8) -- a ray the reference moves by > 1e-4 under a fraction p of the seeds is in the
# union of S seeds with probability 1 - (1 - p)^S; the paired gate (product-bad rays must be reference-unstable rays) needs that union close to complete.
N_SENS_SEEDS = int(os.environ.get("NM_SENS_SEEDS", "32"))
KNOWN = ("scale", "train", "surface", "surf", "surfsens", "surf3sens", "scalesens", "trained", "trainedsens", "reftime", "surf3", "trace", "paint", "edit", "deform", "surface140k", "train140k", "perturb", "edit140k", "trainloop")


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] not in KNOWN:
        raise SystemExit(f"unknown fixture group {sys.argv[1]!r}; one of {KNOWN}, or no argument for all (regenerates the pinned fixtures!)")
    if len(sys.argv) > 1 and sys.argv[1] in KNOWN:   # only one of the later fixtures (the others are unchanged)
        sd = dict(np.load(os.path.join(GOLDEN, "model_seed0.npz")))
        if sys.argv[1] == "scale":
            gen_scale_fixture("render_v140k_dtu", mlp_state=sd)
        elif sys.argv[1] == "surf":   # the scene with a surface (neumesh_amd.synthetic.surface_mlp_state), s = 400
            gen_scale_fixture("render_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0)
        elif sys.argv[1] == "surfsens":   # the reference's own spread over 8 last-bit perturbations of the surf fixture's rays + its speed (reads render_v140k_surf.npz)
            gen_surf_sensitivity("render_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0, n_seeds=N_SENS_SEEDS, timing=False)
        elif sys.argv[1] == "trained":    # round 6: a TRAINED field (tools/train_field.py on the GPU box -> tests/golden/trained_v140k.pt), loaded as render.py:287-288 does
            gen_scale_fixture("render_v140k_trained", ckpt=TRAINED_CKPT)
            gen_surf_sensitivity("render_v140k_trained", n_seeds=N_SENS_SEEDS, timing=False, ckpt=TRAINED_CKPT)
        elif sys.argv[1] == "trainedsens":
            gen_surf_sensitivity("render_v140k_trained", n_seeds=N_SENS_SEEDS, timing=False, ckpt=TRAINED_CKPT)
        elif sys.argv[1] == "surf3sens":  # the same spread for the configs[3]-shape fixture and for the noise-field fixture (round 6: every end-to-end gate is paired)
            gen_surf_sensitivity("render_v140k_surf_c3", mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0, n_seeds=N_SENS_SEEDS, timing=False)
        elif sys.argv[1] == "scalesens":
            gen_surf_sensitivity("render_v140k_dtu", mlp_state=sd, s_value=200.0, n_seeds=N_SENS_SEEDS, timing=False)
        elif sys.argv[1] == "reftime":    # only the timing record of surfsens (REPORT.json "reference_timing"); run it on an otherwise idle machine
            gen_surf_sensitivity("render_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0, n_seeds=0)
        elif sys.argv[1] == "surf3":  # BASELINE configs[3] shape (32 + 32 samples, white background) on the same scene, at headline scale
            gen_scale_fixture("render_v140k_surf_c3", n_rays=1024, mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0,
                              n_samples=32, n_importance=32, white_bkgd=True)
        elif sys.argv[1] == "trace":
            gen_render_py_trace()
        elif sys.argv[1] == "surface140k":
            gen_surface_scale_fixture("surface_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd))
        elif sys.argv[1] == "train140k":
            gen_train_step_fixture("train_step_v140k_surf", V=140_000, mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0, n_rays=256, HW=64,
                                   kdtree=True)
        elif sys.argv[1] == "edit140k":
            gen_texture_edit_scale_fixture("texture_edit_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd))
        elif sys.argv[1] == "perturb":
            gen_perturb_fixture("render_v3000_perturb", V=3000, mlp_state=sd)
        elif sys.argv[1] == "edit":
            gen_texture_edit_fixture("texture_edit_v3000", V=3000, mlp_state=sd)
        elif sys.argv[1] == "deform":
            gen_deform_fixture("deform_v3000", V=3000, mlp_state=sd)
        elif sys.argv[1] == "paint":
            gen_painting_step_fixture("painting_step_v3000", V=3000, mlp_state=sd)
        elif sys.argv[1] == "train":
            gen_train_step_fixture("train_step_v3000", V=3000, mlp_state=sd)
        elif sys.argv[1] == "trainloop":
            gen_train_loop_fixture("train_loop_v3000", V=3000, mlp_state=sd)
        else:
            gen_surface_fixture("surface_v3000", V=3000, mlp_state=sd)
        rp = os.path.join(GOLDEN, "REPORT.json")
        old = json.load(open(rp)) if os.path.exists(rp) else {}
        old.update(REPORT)
        with open(rp, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
        return
    gen_rays_fixture()
    model = gen_field_fixture("field_v3000", V=3000, Q=2048, seed=11)
    # one weights file shared by every fixture (reference ctor, torch.manual_seed(0))
    sd = {k: v.detach().numpy() for k, v in model.state_dict().items()
          if k not in ("geometry_features", "color_features", "indicator_vector")}
    np.savez_compressed(os.path.join(GOLDEN, "model_seed0.npz"), **sd)
    gen_field_fixture("field_dup_v1200", V=1200, Q=512, seed=12, dup=64, mlp_state=sd)
    gen_render_fixture("render_v3000_dtu", V=3000, H=6, W=12, frame=3, mlp_state=sd)
    gen_render_fixture("render_v3000_lego", V=3000, H=4, W=12, frame=17, white_bkgd=True, n_samples=32, mlp_state=sd)
    gen_grad_fixture("grad_v3000_dtu", "render_v3000_dtu", V=3000, mlp_state=sd)
    gen_train_step_fixture("train_step_v3000", V=3000, mlp_state=sd)
    gen_surface_fixture("surface_v3000", V=3000, mlp_state=sd)
    gen_scale_fixture("render_v140k_dtu", mlp_state=sd)
    gen_scale_fixture("render_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0)
    gen_scale_fixture("render_v140k_surf_c3", n_rays=1024, mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0,
                      n_samples=32, n_importance=32, white_bkgd=True)
    gen_render_py_trace()
    gen_painting_step_fixture("painting_step_v3000", V=3000, mlp_state=sd)
    gen_texture_edit_fixture("texture_edit_v3000", V=3000, mlp_state=sd)
    gen_deform_fixture("deform_v3000", V=3000, mlp_state=sd)
    gen_perturb_fixture("render_v3000_perturb", V=3000, mlp_state=sd)
    gen_texture_edit_scale_fixture("texture_edit_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd))
    gen_surface_scale_fixture("surface_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd))
    gen_train_step_fixture("train_step_v140k_surf", V=140_000, mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0, n_rays=256, HW=64, kdtree=True)
    gen_surf_sensitivity("render_v140k_surf", mlp_state=synthetic.surface_mlp_state(sd), s_value=400.0)
    with open(os.path.join(GOLDEN, "REPORT.json"), "w") as f:
        json.dump(REPORT, f, indent=1, sort_keys=True)
    print("all oracle-vs-reference checks passed; fixtures written to", GOLDEN)


if __name__ == "__main__":
    main()
