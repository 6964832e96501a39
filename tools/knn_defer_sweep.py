"""tools/knn_defer_sweep.py -- GPU box: the K-NN launches of a training batch (512 random rays of the bench frame: 256 probes, 64 coarse
samples per ray) timed for several work budgets of the traversal (NEUMESH_KNN_BUDGET; 0 = no deferral, see csrc/nm_api.hip)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NEUMESH_HIP_LIB"] = os.path.join(ROOT, "tests", "_build", "libneumesh_hip_testing.so")   # (for nm_debug_last_deferred)
import ctypes as C
import torch, bench
from neumesh_amd import synthetic, rays as R, _lib
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
H = W = 800
pose, K = synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W)
o, d = R.make_rays(pose, K, H, W, dev)
sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:512].to(dev)
o, d = o[sel], torch.nn.functional.normalize(d[sel], dim=-1)
b = (o * d).sum(-1)
disc = (b * b - ((o * o).sum(-1) - 1.0)).clamp_min(0).sqrt()
near, far = (-b - disc).clamp_min(0.0), (-b + disc)
tt = torch.linspace(0, 1, 256, device=dev)
probes = (o[:, None, :] + (near[:, None] + (far - near)[:, None] * tt[None, :])[..., None] * d[:, None, :]).reshape(-1, 3).contiguous()
t64 = torch.linspace(0.25, 0.75, 128, device=dev)
dense = (o[:, None, :] + (near[:, None] + (far - near)[:, None] * t64[None, :])[..., None] * d[:, None, :]).reshape(-1, 3).contiguous()
small = dense.reshape(512, 128, 3)[:, ::8].reshape(-1, 3).contiguous()


def timeit(pts, n=10):
    with torch.no_grad():
        ref = model.compute_distance(pts)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            model.compute_distance(pts)
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, ref


budgets = [int(x) for x in sys.argv[1:]] or [0, 15000, 20000, 30000, 45000, 60000, 100000]
for name, pts in (("probes 512 x 256", probes), ("samples 512 x 128", dense), ("samples 512 x 16", small)):
    base = None
    for bdg in budgets:
        os.environ["NEUMESH_KNN_BUDGET"] = str(bdg)
        us, out = timeit(pts)
        if base is None:
            base = out
        same = all(torch.equal(a, b) for a, b in zip(out, base))
        cnt = C.c_int(-1)
        lib = _lib.load()
        fn = lib.nm_debug_last_deferred
        fn.restype, fn.argtypes = C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]
        fn(model.mesh_grid.grid.handle, C.byref(cnt), _lib.current_stream(dev))
        print(f"{name:20s} {pts.shape[0]:7d} points  budget {bdg:7d}: {us:8.0f} us per call   deferred {cnt.value:6d}   identical to budget 0: {same}", flush=True)
