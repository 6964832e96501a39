"""tools/knn_hard_split.py -- GPU box: would a training step's point-wise K-NN launches gain from searching the 'hard' queries (near the centre of
the object, where nearly every leaf has to be scanned) in their own launch, one query per wave?  Times compute_distance on the sample points of a
512-ray batch: all points in one launch (what a step does), the points outside |x| < rho and the points inside it as two launches."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neumesh_amd import synthetic
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
H = W = 800
o, d = synthetic.camera_rays(synthetic.orbit_pose(3), synthetic.pinhole_intrinsics(H, W), H, W)
rng = np.random.default_rng(0)
sel = rng.integers(0, H * W, 512)
o, d = torch.from_numpy(o[sel]).to(dev), torch.from_numpy(d[sel]).to(dev)
d = d / d.norm(dim=-1, keepdim=True)
mid = -(o * d).sum(-1, keepdim=True)
for name, n in (("probes (256 per ray)", 256), ("samples (128 per ray)", 128), ("coarse (64 per ray)", 64)):
    t = torch.linspace(0, 1, n, device=dev)
    near, far = (mid - 1.0).clamp_min(0), mid + 1.0
    dep = near * (1 - t) + far * t
    pts = (o[:, None, :] + dep[..., None] * d[:, None, :]).reshape(-1, 3).contiguous()
    r = pts.norm(dim=-1)

    def run(p, reps=20):
        with torch.no_grad():
            for _ in range(3):
                model.compute_distance(p)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                model.compute_distance(p)
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    line = f"{name}: {pts.shape[0]} points, all in one launch {run(pts):.3f} ms"
    for rho in (0.25, 0.4, 0.55):
        hard = r < rho
        a, b = pts[~hard].contiguous(), pts[hard].contiguous()
        line += f" | rho {rho}: {int(hard.sum())} hard: easy {run(a):.3f} + hard {run(b):.3f} ms"
    print(line, flush=True)
