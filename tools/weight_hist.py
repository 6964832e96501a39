"""tools/weight_hist.py -- GPU box: distribution of the visibility weights of the bench frame's mid-points
(how many are exactly 0, how many are positive but below 1e-9 / 1e-7 / 1e-5)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neumesh_amd import synthetic
from neumesh_amd.rays import make_rays
from neumesh_amd.renderer import volume_render
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
H = W = 800
ro, rd = make_rays(synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W), H, W, dev)
sel = torch.arange(0, H * W, 5, device=dev)
with torch.no_grad():
    rgb, depth, ret = volume_render(ro[sel][None], rd[sel][None], model, calc_normal=True, perturb=False, detailed_output=True,
                                    N_samples=64, N_importance=64, bounded_near_far=True, batched=True, rayschunk=32768, obj_bounding_radius=1.0)
w = ret["visibility_weights"].flatten().double().cpu().numpy()
n = w.size
print(f"{n} mid-points of {len(sel)} rays: w == 0: {np.mean(w == 0):.3f}")
for eps in (1e-12, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
    m = (w > 0) & (w < eps)
    wr = ret["visibility_weights"][0].double().cpu().numpy()
    per_ray = np.where((wr > 0) & (wr < eps), wr, 0).sum(-1)
    print(f"  0 < w < {eps:g}: {m.mean():.3f} of all = {m.sum() / max((w > 0).sum(), 1):.3f} of the evaluated ones; largest dropped weight sum of a ray {per_ray.max():.2e}")
