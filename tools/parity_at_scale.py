"""tools/parity_at_scale.py -- diagnostic (GPU box): where do GPU and oracle renders differ on the
V=1.4e5 benchmark scene, and is it the field or the sample placement?"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neumesh_amd.renderer import volume_render
from oracle import compare, field as ofield, knn as oknn, render as orender

dev = torch.device("cuda", 0)
H = W = 800
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 384
mesh, model = bench.build_scene(140000, dev)
o, d = bench.frame_rays(0, H, W)
sel = np.linspace(0, H * W - 1, n_rays).astype(np.int64)
o, d = o[sel], d[sel]
with torch.no_grad():
    rgb, depth, ex = volume_render(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), model, calc_normal=True,
                                   perturb=False, detailed_output=True, rayschunk=65536)
g = {k: v.cpu().numpy() for k, v in ex.items()}
state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
orc = ofield.OracleField(mesh.vertices, state, ofield.FieldConfig(speed_factor=10.0))   # brute-force K-NN (exact declaration)
out = orender.render_rays(orc, o, d, orender.RenderConfig(calc_normal=True), detailed=True)
err = np.abs(g["rgb"] - out["rgb"]).max(-1)
print("rgb err quantiles (50/90/99/max):", np.quantile(err, [0.5, 0.9, 0.99, 1.0]))
print("rays > 1e-4:", int((err > 1e-4).sum()), "of", n_rays, " psnr", compare.psnr(g["rgb"], out["rgb"]))
print("near/far max diff:", np.abs(g["near_far"] - np.concatenate([out["near"], out["far"]], 1)).max())
same_d = np.all(g["d_all"] == out["d_all"], axis=1)
print("rays with bit-identical d_all:", int(same_d.sum()), " max err among them:", float(err[same_d].max()) if same_d.any() else None,
      " max err among the others:", float(err[~same_d].max()) if (~same_d).any() else None)
# field parity on the ORACLE's own sample points (no sampling differences involved)
dn = orender.normalize(d)
pts = (o[:, None, :] + dn[:, None, :] * out["d_all"][..., None]).astype(np.float32)
with torch.no_grad():
    sdf_g, nab_g = model.forward_with_nablas(torch.from_numpy(pts).to(dev))
print("field on oracle's points: max |sdf| err", float(np.abs(sdf_g[..., 0].cpu().numpy() - out["implicit_surface"]).max()),
      " nabla err", float(np.abs(nab_g.cpu().numpy() - out["implicit_nablas"]).max()))
# conditioning of the reference itself: oracle re-rendered with every ray direction nudged by 1 ulp
d2 = np.nextafter(d, np.float32(10), dtype=np.float32)
out2 = orender.render_rays(orc, o, d2, orender.RenderConfig(calc_normal=True))
e2 = np.abs(out2["rgb"] - out["rgb"]).max(-1)
print("oracle vs oracle with ray directions nudged by 1 ulp: quantiles", np.quantile(e2, [0.5, 0.9, 0.99, 1.0]), " rays > 1e-4:", int((e2 > 1e-4).sum()))
worst = np.argsort(-err)[:5]
for r in worst:
    ws, frac = compare.depth_set_distance(g["d_all"][r:r + 1], out["d_all"][r:r + 1])
    print(f" ray {r}: err {err[r]:.2e} acc {out['mask_volume'][r]:.3f} d_all set-dist {ws:.2e} unmatched {frac:.3f} near/far {out['near'][r,0]:.3f}/{out['far'][r,0]:.3f}")
