"""tools/stress5_split.py -- debug (GPU box): how the config-5 kernel's time splits between the K-NN search and
the 8-row gather: same 2^20 coherent queries with a 256-d, a 32-d and a 4-d table."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neumesh_amd import synthetic
from neumesh_amd.mesh_grid import MeshGrid
from neumesh_amd.rays import make_rays
dev = torch.device("cuda", 0)
V, H, W = 1_000_000, 4096, 4096
mesh = synthetic.fibonacci_blob(V)
grid = MeshGrid(bench._Mesh(mesh), dev)
o, d = make_rays(synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W), H, W, dev, first_pixel=1920 * W, count=256 * W)
d = torch.nn.functional.normalize(d, dim=-1)
b = (o * d).sum(-1)
t = -b - torch.sqrt(torch.clamp(b * b - ((o * o).sum(-1) - 0.75 ** 2), min=0.0))
q = (o + t[:, None] * d).contiguous()
print("leaf level", grid.grid.info() if hasattr(grid.grid, "info") else "")
for dim in (256, 32, 4):
    table = torch.randn((V, dim), device=dev)
    with torch.no_grad():
        for _ in range(2):
            grid.compute_distance_interpolate(q, table)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            grid.compute_distance_interpolate(q, table)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"dim {dim:4d}: {dt * 1e3:7.3f} ms per 2^20 queries  ({q.shape[0] / dt / 1e6:.0f} Mq/s)")
