import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import bench
from neumesh_amd import synthetic, _lib
from neumesh_amd.rays import make_rays
from neumesh_amd.renderer import make_render_cfg, render_rays_fused
dev = torch.device("cuda", 0)
lib = _lib.load()
mesh, model = bench.build_scene(140_000, dev, scene="surf")
H = W = 800
intr = synthetic.pinhole_intrinsics(H, W)
rays = [make_rays(synthetic.orbit_pose(s), intr, H, W, dev) for s in range(8)]
cfg = make_render_cfg(calc_normal=True, N_samples=64, N_importance=64)
tables = model.field_tables(); model.field_handle()
prof = int(os.environ.get("DIAG_PROF", "1"))
out = []
for i in range(8):
    if i == 1 and prof: lib.nm_profile_enable(1)
    t0 = time.perf_counter()
    ret = render_rays_fused(model, rays[i][0], rays[i][1], cfg, H * W, tables=tables)
    t1 = time.perf_counter()
    if i == 0 and os.environ.get("DIAG_CPU"):
        _ = ret["rgb"].cpu().numpy()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    out.append(f"{(t1-t0)*1e3:.0f}+{(t2-t1)*1e3:.0f}")
print("prof", prof, "keep", os.environ.get("NEUMESH_WS_KEEP_GB"), "enqueue+drain ms per frame:", " ".join(out), flush=True)
