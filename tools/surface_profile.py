"""tools/surface_profile.py -- GPU box: the surface renderer's frame on the bench scene (for rocprofv3 --kernel-trace --stats)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from neumesh_amd import ray_casting as rc
dev = torch.device("cuda", 0)
scene = sys.argv[1] if len(sys.argv) > 1 else "surf"
mesh, model = bench.build_scene(140000, dev, scene=scene)
o, d = bench.frame_rays(0, 800, 800)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
with torch.no_grad():
    tau = float(model.forward_density_only(torch.from_numpy(mesh.vertices[::7].astype(np.float32)).to(dev)).median())
    cfgs = dict(near=0.5, far=3.5, logit_tau=tau, fill_inf=False)
    for i in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        out = rc.surface_render(ro[None], rd[None], model, calc_normal=True, batched=True, rayschunk=1 << 17, ray_casting_algo="root_finding", ray_casting_cfgs=cfgs)
        torch.cuda.synchronize(); print(scene, "frame %.1f ms, hit %.3f" % ((time.perf_counter() - t) * 1e3, float(out[2]["mask_surface"].float().mean())))
