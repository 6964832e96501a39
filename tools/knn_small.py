"""tools/knn_small.py -- GPU box: K-NN + distance kernel time of small point sets (a training step's launches) in different point orders.
(NEUMESH_KNN_LANES=n forces the number of queries per wave: honoured by the -DNM_TESTING build only -- NEUMESH_HIP_LIB=tests/_build/libneumesh_hip_testing.so.)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from neumesh_amd import synthetic, rays as R
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
H = W = 800
pose, K = synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W)
o, d = R.make_rays(pose, K, H, W, dev)
g = torch.Generator(device="cpu").manual_seed(0)
sel = torch.randperm(H * W, generator=g)[:512].to(dev)
o, d = o[sel], torch.nn.functional.normalize(d[sel], dim=-1)
near = (-(o * d).sum(-1) - 1.0).clamp_min(0.05)

def pts(n, jitter):
    t = torch.linspace(0, 1, n, device=dev)[None, :] * 2.0 + near[:, None]
    if jitter:
        t = t + torch.rand(t.shape, device=dev, generator=None) * (2.0 / n)
    return o[:, None, :] + t[..., None] * d[:, None, :]

def morton(p):
    q = ((p * 0.5 + 0.5).clamp(0, 1) * 1023).to(torch.int64)
    for sh, msk in ((16, 0x30000FF), (8, 0x300F00F), (4, 0x30C30C3), (2, 0x9249249)):
        q = (q | (q << sh)) & msk
    return q[:, 0] | (q[:, 1] << 1) | (q[:, 2] << 2)

def timed(x, reps=5):
    x = x.contiguous()
    model.compute_distance(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.compute_distance(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

with torch.no_grad():
    for n in (16, 64, 128, 256):
        p = pts(n, True)
        flat = p.reshape(-1, 3)
        tile = model._tile_order(p.shape, dev)
        a = timed(flat)
        b = timed(flat[tile[0]]) if tile is not None else float("nan")
        perm = torch.argsort(morton(flat))
        c = timed(flat[perm])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            perm = torch.argsort(morton(flat)); x = flat[perm]
        torch.cuda.synchronize(); s = (time.perf_counter() - t0) / 5 * 1e3
        print(f"512 rays x {n:3d} samples = {flat.shape[0]:6d} points: ray-major {a:.3f} ms, 16x4 tiles {b:.3f} ms, Morton-sorted points {c:.3f} ms (+ sort by torch ops {s:.3f} ms)", flush=True)

# where does the ~1 ms of a small launch come from?  near-surface / far points, sorted / random, several sizes
V = torch.from_numpy(np.asarray(mesh.vertices, np.float32)).to(dev)
with torch.no_grad():
    for n in (512, 8192, 65536):
        gidx = torch.randint(0, V.shape[0], (n,), device=dev)
        nearp = V[gidx] + 0.01 * torch.randn(n, 3, device=dev)
        farp = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=-1) * 0.95
        for name, p in (("near surface", nearp), ("far shell r=0.95", farp)):
            a = timed(p)
            c = timed(p[torch.argsort(morton(p))])
            print(f"{n:6d} points {name:18s}: random order {a:.3f} ms, Morton order {c:.3f} ms", flush=True)
