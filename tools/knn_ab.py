"""tools/knn_ab.py -- GPU box: plain K-NN kernel (search only) vs the fused K-NN + distance (+ gather) kernel on the
same ray-ordered query points; answers how much of the fused kernel is traversal and how much epilogue."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neumesh_amd import _lib, synthetic
from neumesh_amd.mesh_grid import knn
dev = torch.device("cuda", 0)
lib = _lib.load()
mesh, model = bench.build_scene(140000, dev)
o, d = bench.frame_rays(0, 800, 800)
sel = np.arange(0, 640000, 5)           # 128 000 rays, 64 samples each in [1.35, 2.3] along the ray: 8.2 M points
t = np.linspace(1.35, 2.3, 64, dtype=np.float32)
pts = (o[sel][:, None, :] + d[sel][:, None, :] * t[None, :, None]).reshape(-1, 3).astype(np.float32)
x = torch.from_numpy(pts).to(dev)
P = x.shape[0]
grid = model.mesh_grid

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

idx = torch.empty((P, 8), dtype=torch.int64, device=dev); d2 = torch.empty((P, 8), device=dev)
st = _lib.current_stream(dev)
ms_knn = timeit(lambda: lib.nm_knn(grid.grid.handle, _lib.ptr(x), P, 8, _lib.ptr(idx), _lib.ptr(d2), st))
ds = torch.empty((P,), device=dev); gr = torch.empty((P, 3), device=dev)
ind = model.indicator_vector.detach().contiguous()
ms_dist = timeit(lambda: lib.nm_compute_distance(grid.grid.handle, _lib.ptr(x), P, _lib.ptr(ind), 0.1, 8, _lib.ptr(ds), None, None, _lib.ptr(gr), st))
feat = torch.empty((P, 32), device=dev)
tab = model.geometry_features.detach().contiguous()
ms_gather = timeit(lambda: lib.nm_distance_interpolate(grid.grid.handle, _lib.ptr(x), P, _lib.ptr(ind), 0.1, _lib.ptr(tab), 32, _lib.ptr(ds), None, None, _lib.ptr(feat), st))
print(f"lib {os.environ.get('NEUMESH_HIP_LIB', 'default')}: {P} points: plain K-NN (idx64 + d2 out) {ms_knn:.2f} ms = {P / ms_knn / 1e6:.2f} Gq/s; "
      f"K-NN + distance + grad {ms_dist:.2f} ms; K-NN + distance + 32-d gather {ms_gather:.2f} ms")
