"""tools/v3_check.py -- GPU box: f16x2 vs f16x2_v3 on the bench scene: field calls on 2^18 points and one rendered ray set."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench, common
from neumesh_amd.renderer import volume_render
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
rng = np.random.default_rng(1)
for P in (1000, 1 << 18):
    x = torch.from_numpy((mesh.vertices[rng.integers(0, 140000, P)] + 0.02 * rng.standard_normal((P, 3))).astype(np.float32)).to(dev)
    v = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
    out = {}
    for mode in ("f16x2", "f16x2_v3"):
        model.mlp_precision = mode
        with torch.no_grad():
            s0 = model.forward_density_only(x)
            s1, nab = model.forward_with_nablas(x)
            s2, rgb = model.forward(x, v)
        out[mode] = (s0, s1, nab, rgb)
        print(P, mode, "fwd==nabla values:", bool(torch.equal(s0, s1)), bool(torch.equal(s0, s2)))
    a, b = out["f16x2"], out["f16x2_v3"]
    for name, i in (("sdf", 0), ("nabla", 2), ("rgb", 3)):
        d = (a[i] - b[i]).abs()
        print(P, name, "max |v2 - v3| =", float(d.max()), "at", int(d.flatten().argmax()) // max(1, d[0].numel()))
f = common.golden("render_v140k_dtu")
ro, rd = torch.from_numpy(f["rays_o"]).to(dev), torch.from_numpy(f["rays_d"]).to(dev)
for n in (64, 128, 200, 1536):
    res = {}
    for mode in ("f16x2", "f16x2_v3"):
        model.mlp_precision = mode
        with torch.no_grad():
            rgb, depth, ex = volume_render(ro[:n], rd[:n], model, calc_normal=True, N_samples=64, N_importance=64, perturb=False, rayschunk=65536, detailed_output=True)
        res[mode] = ex
    for k in ("rgb", "implicit_surface", "implicit_nablas", "radiance", "d_final"):
        d = (res["f16x2"][k] - res["f16x2_v3"][k]).abs()
        print(n, k, "max diff", float(d.max()), "median", float(d.flatten().median()))
print("production path (detailed_output=False), v3 vs v2, by evaluation-strategy flags:")
for env in ("", "NEUMESH_EAGER_NABLAS", "NEUMESH_NO_ZERO_SKIP", "NEUMESH_NO_MID_ORDER", "NEUMESH_NO_RAY_SORT"):
    for e in ("NEUMESH_EAGER_NABLAS", "NEUMESH_NO_ZERO_SKIP", "NEUMESH_NO_MID_ORDER", "NEUMESH_NO_RAY_SORT"):
        os.environ.pop(e, None)
    if env: os.environ[env] = "1"
    for n in (72, 1536):
        res = {}
        for mode in ("f16x2", "f16x2_v3"):
            model.mlp_precision = mode
            with torch.no_grad():
                rgb, depth, ex = volume_render(ro[:n], rd[:n], model, calc_normal=True, N_samples=64, N_importance=64, perturb=False, rayschunk=65536, detailed_output=False)
            res[mode] = (rgb, depth, ex["normals_volume"], ex["mask_volume"])
        d = [float((a - b).abs().max()) for a, b in zip(res["f16x2"], res["f16x2_v3"])]
        bad = int(((res["f16x2"][0] - res["f16x2_v3"][0]).abs().amax(-1) > 1e-4).sum())
        print(f"{env or 'default':22s} n={n}: max diff rgb {d[0]:.2e} depth {d[1]:.2e} normals {d[2]:.2e} acc {d[3]:.2e}; rays > 1e-4: {bad}")
