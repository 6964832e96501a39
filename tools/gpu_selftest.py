"""tools/gpu_selftest.py -- one-shot on-device diagnostic (run on the GPU box):

    python tools/gpu_selftest.py [--out gpurun_out/selftest.json]

Runs every stage of the HIP path against the golden fixtures / the oracle and prints compact
error statistics (never stops at the first failure), so one GPU session yields as much
information as possible.  Not part of the product; the pass/fail gates live in tests/ (-m gpu).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

RESULTS = {}


def section(name):
    def deco(fn):
        def run(*a, **k):
            t = time.time()
            try:
                out = fn(*a, **k)
                RESULTS[name] = {"ok": True, "secs": round(time.time() - t, 3), **(out or {})}
            except Exception as e:  # noqa: BLE001
                RESULTS[name] = {"ok": False, "error": f"{type(e).__name__}: {e}", "trace": traceback.format_exc()[-1500:]}
            print(f"[{name}] {json.dumps(RESULTS[name])[:1200]}", flush=True)
        return run
    return deco


def stats(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    d = np.abs(got - want)
    return {"max": float(d.max()), "mean": float(d.mean()), "scale": float(np.abs(want).max()),
            "nan": int(np.isnan(got).sum())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "selftest.json"))
    args = ap.parse_args()
    import torch
    import common
    from neumesh_amd import _lib
    from oracle import knn as oknn, render as orender, compare

    dev = torch.device("cuda", 0)
    print("device:", torch.cuda.get_device_name(0), "lib devices:", _lib.load().nm_device_count(), flush=True)

    fx = common.golden("field_v3000")
    mesh = common.scene_mesh(3000)
    state = common.scene_state(mesh)
    model = common.make_model(mesh, state, dev)
    print("grid:", model.mesh_grid.grid.info(), flush=True)
    q = torch.from_numpy(fx["q"]).to(dev)
    dirs = torch.from_numpy(fx["dirs"]).to(dev)

    @section("knn_vs_golden")
    def t_knn():
        from neumesh_amd.mesh_grid import knn
        idx, d2 = knn(model.mesh_grid.grid, q, 8)
        torch.cuda.synchronize()
        idx, d2 = idx.cpu().numpy(), d2.cpu().numpy()
        return {"idx_mismatch_rows": int((idx != fx["idx"]).any(1).sum()), "d2_bit_equal": bool(np.array_equal(d2, fx["d2"])),
                "rows": int(idx.shape[0])}
    t_knn()

    @section("knn_other_K")
    def t_knn_k():
        from neumesh_amd.mesh_grid import knn
        out = {}
        for K in (1, 3, 8, 16, 32):
            idx, d2 = knn(model.mesh_grid.grid, q[:512], K)
            ri, rd = oknn.knn_bruteforce(fx["q"][:512], mesh.vertices, K)
            out[f"K{K}"] = bool(np.array_equal(idx.cpu().numpy(), ri) and np.array_equal(d2.cpu().numpy(), rd))
        return out
    t_knn_k()

    @section("knn_dup_ties")
    def t_dup():
        fxd = common.golden("field_dup_v1200")
        meshd = common.scene_mesh(1200, 64)
        md = common.make_model(meshd, common.scene_state(meshd), dev)
        from neumesh_amd.mesh_grid import knn
        idx, d2 = knn(md.mesh_grid.grid, torch.from_numpy(fxd["q"]).to(dev), 8)
        with torch.no_grad():
            sdf, nab = md.forward_with_nablas(torch.from_numpy(fxd["q"]).to(dev))
        return {"idx_mismatch_rows": int((idx.cpu().numpy() != fxd["idx"]).any(1).sum()),
                "d2_bit_equal": bool(np.array_equal(d2.cpu().numpy(), fxd["d2"])),
                "sdf": stats(sdf.cpu().numpy(), fxd["sdf"])}
    t_dup()

    @section("compute_distance")
    def t_dist():
        with torch.no_grad():
            ds, idx, w = model.compute_distance(q)
            ds2, idx2, w2, g = model.mesh_grid.compute_distance_frnn(q, 8, model.indicator_vector, 0.1, want_grad=True)
        return {"ds": stats(ds.cpu().numpy(), fx["ds"]), "w": stats(w.cpu().numpy(), fx["w"]),
                "idx_equal": bool(np.array_equal(idx.cpu().numpy(), fx["idx"])), "idx_dtype": str(idx.dtype),
                "dds_dx": stats(g.cpu().numpy(), fx["dds_dx"])}
    t_dist()

    near = np.abs(fx["ds"][:, 0]) < 0.2

    @section("field_density")
    def t_den():
        with torch.no_grad():
            sdf = model.forward_density_only(q)
            sdf2, nab = model.forward_with_nablas(q)
        sdf, sdf2, nab = sdf.cpu().numpy(), sdf2.cpu().numpy(), nab.cpu().numpy()
        tol = 5e-6 + 2e-4 * np.abs(fx["ds"])
        return {"sdf": stats(sdf, fx["sdf"]), "sdf_with_nabla": stats(sdf2, fx["sdf"]),
                "sdf_paths_bit_equal": bool(np.array_equal(sdf, sdf2)),
                "nabla_all": stats(nab, fx["nabla"]), "nabla_near": stats(nab[near], fx["nabla"][near]),
                "nabla_within_tol": bool(np.all(np.abs(nab - fx["nabla"]) <= tol))}
    t_den()

    @section("field_forward")
    def t_fwd():
        with torch.no_grad():
            sdf, rgb, ds, idx, w = model.forward(q, dirs, return_ds=True)
            sdf_n, nab = model.forward(q, dirs, nablas_only=True)
            rgb2 = model.forward_color(ds, dirs, model.color_features, idx, w, nab)
        return {"sdf": stats(sdf.cpu().numpy(), fx["sdf"]), "rgb": stats(rgb.cpu().numpy(), fx["rgb"]),
                "rgb_forward_color": stats(rgb2.cpu().numpy(), fx["rgb"]),
                "ds": stats(ds.cpu().numpy(), fx["ds"]), "idx_equal": bool(np.array_equal(idx.cpu().numpy(), fx["idx"]))}
    t_fwd()

    @section("mfma_vs_valu_selfcheck")
    def t_self():
        import ctypes as C
        lib = _lib.load_testing()   # nm_selfcheck_field: test hook of the -DNM_TESTING library
        P = q.shape[0]
        sdf = torch.empty((P,), device=dev); nab = torch.empty((P, 3), device=dev); rgb = torch.empty((P, 3), device=dev)
        scratch = torch.empty((int(lib.nm_field_scratch_bytes(P)),), dtype=torch.uint8, device=dev)
        tmp = torch.empty(((P + 31) // 32 * 64 * 256,), device=dev)
        t, keep = model.field_tables()
        _lib.check(lib.nm_selfcheck_field(model.field_handle(), model.mesh_grid.grid.handle, C.byref(t), _lib.ptr(q), _lib.ptr(dirs), P,
                                          _lib.ptr(sdf), _lib.ptr(nab), _lib.ptr(rgb), _lib.ptr(scratch), _lib.ptr(tmp),
                                          _lib.current_stream(dev)), "selfcheck")
        torch.cuda.synchronize()
        return {"valu_sdf_vs_golden": stats(sdf.cpu().numpy(), fx["sdf"][:, 0]), "valu_rgb_vs_golden": stats(rgb.cpu().numpy(), fx["rgb"]),
                "valu_nabla_near": stats(nab.cpu().numpy()[near], fx["nabla"][near])}
    t_self()

    @section("autograd_path")
    def t_auto():
        qq = q[:256].clone()
        sdf, nab = model.forward_with_nablas(qq)
        sdf2, rgb = model.forward(q[:256].clone(), dirs[:256])
        return {"sdf": stats(sdf.detach().cpu().numpy(), fx["sdf"][:256]), "nabla_near": stats(
            nab.detach().cpu().numpy()[near[:256]], fx["nabla"][:256][near[:256]]), "rgb": stats(rgb.detach().cpu().numpy(), fx["rgb"][:256])}
    t_auto()

    for tag in ("render_v3000_dtu", "render_v3000_lego"):
        @section(tag)
        def t_render(tag=tag):
            rf = common.golden(tag)
            from neumesh_amd.renderer import volume_render
            ro, rd = torch.from_numpy(rf["rays_o"]).to(dev), torch.from_numpy(rf["rays_d"]).to(dev)
            ns = int(rf["N_samples"])
            with torch.no_grad():
                rgb, depth, ex = volume_render(ro[None], rd[None], model, batched=True, calc_normal=bool(rf["calc_normal"]),
                                               white_bkgd=bool(rf["white_bkgd"]), N_samples=ns, N_importance=ns, rayschunk=4096,
                                               detailed_output=True, perturb=False)
            ex = {k: v[0].cpu().numpy() for k, v in ex.items()}
            worst, far = compare.depth_set_distance(ex["d_all"], rf["d_all"])
            out = {"rgb": stats(ex["rgb"], rf["rgb"]), "depth": stats(ex["depth_volume"], rf["depth_volume"]),
                   "acc": stats(ex["mask_volume"], rf["mask_volume"]), "near_far": stats(ex["near_far"], np.concatenate([rf["near"], rf["far"]], 1)),
                   "d_all_set_max": worst, "d_all_unmatched_frac": far, "psnr": compare.psnr(ex["rgb"], rf["rgb"])}
            if "normals_volume" in ex:
                out["normals"] = stats(ex["normals_volume"], rf["normals_volume"])
            return out
        t_render()

    @section("timing_small")
    def t_time():
        import ctypes as C
        lib = _lib.load()
        big = common.scene_mesh(140000)
        st = common.scene_state(big)
        mb = common.make_model(big, st, dev)
        P = 1 << 20
        rng = np.random.default_rng(0)
        pts = big.vertices[rng.integers(0, 140000, P)] + 0.02 * rng.standard_normal((P, 3)).astype(np.float32)
        x = torch.from_numpy(pts.astype(np.float32)).to(dev)
        v = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
        scratch = torch.empty((int(lib.nm_field_scratch_bytes(P)),), dtype=torch.uint8, device=dev)
        t, keep = mb.field_tables()
        out = {"grid": mb.mesh_grid.grid.info()}
        flops = {1: 353280, 2: 353280 + 271360, 3: 500736}
        for which, name in ((0, "knn_distance"), (1, "geo_mlp"), (2, "geo_mlp_nabla"), (3, "col_mlp")):
            ms = C.c_float()
            _lib.check(lib.nm_time_kernel(mb.field_handle(), mb.mesh_grid.grid.handle, C.byref(t), which, _lib.ptr(x), _lib.ptr(v), P,
                                          _lib.ptr(scratch), 3, C.byref(ms), _lib.current_stream(dev)), name)
            out[name + "_ms"] = round(ms.value, 3)
            out[name + "_Mpts_s"] = round(P / ms.value / 1e3, 1)
            if which in flops:
                out[name + "_TFLOPs"] = round(P * flops[which] / ms.value / 1e9, 2)
        # far queries (the 256-probe regime)
        far = torch.from_numpy(rng.uniform(-1, 1, (P, 3)).astype(np.float32)).to(dev)
        ms = C.c_float()
        _lib.check(lib.nm_time_kernel(mb.field_handle(), mb.mesh_grid.grid.handle, C.byref(t), 0, _lib.ptr(far), _lib.ptr(v), P,
                                      _lib.ptr(scratch), 3, C.byref(ms), _lib.current_stream(dev)), "far")
        out["knn_distance_uniform_cube_ms"] = round(ms.value, 3)
        return out
    t_time()

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(RESULTS, f, indent=1)
    bad = [k for k, v in RESULTS.items() if not v.get("ok")]
    print("sections failed:", bad)


if __name__ == "__main__":
    main()
