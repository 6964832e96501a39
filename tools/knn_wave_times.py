"""tools/knn_wave_times.py -- GPU box, -DNM_TESTING library: per-wave life of the distance kernel in the K-NN launches of a training step
(512 random rays of the bench frame; probes, coarse samples, importance samples).  Why does a 32 k-point launch take 1 ms?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NEUMESH_HIP_LIB"] = os.path.join(ROOT, "tests", "_build", "libneumesh_hip_testing.so")
import ctypes as C
import torch, bench
from neumesh_amd import synthetic, rays as R, _lib
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
lib = _lib.load()
H = W = 800
pose, K = synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W)
o, d = R.make_rays(pose, K, H, W, dev)
sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:512].to(dev)
o, d = o[sel], torch.nn.functional.normalize(d[sel], dim=-1)
b = (o * d).sum(-1)
disc = (b * b - ((o * o).sum(-1) - 1.0)).clamp_min(0).sqrt()
near, far = (-b - disc).clamp_min(0.0), (-b + disc)
log = torch.zeros(1 + 3 * (1 << 20), dtype=torch.int64, device=dev)

def run(name, pts):
    flat = pts.reshape(-1, 3).contiguous()
    with torch.no_grad():
        model.compute_distance(flat)                      # warm
        log.zero_()
        _lib.check(lib.nm_debug_wave_log(_lib.ptr(log)), "log")
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        ds = model.compute_distance(flat)[0]
        ev1.record()
        torch.cuda.synchronize()
        _lib.check(lib.nm_debug_wave_log(None), "log")
    n = int(log[0])
    t = log[1:1 + 3 * n].reshape(n, 3).cpu().numpy()
    dur = (t[:, 1] - t[:, 0]) / 100.0                     # s_memrealtime: 100 MHz -> microseconds
    t0 = t[:, 0].min()
    span = (t[:, 1].max() - t0) / 100.0
    start = (t[:, 0] - t0) / 100.0
    q = np.percentile(dur, [50, 90, 99, 100])
    print(f"{name:34s} {flat.shape[0]:7d} pts {n:6d} waves  call {ev0.elapsed_time(ev1) * 1e3:7.0f} us  kernel span {span:7.0f} us  wave life p50 {q[0]:6.0f} p90 {q[1]:6.0f} "
          f"p99 {q[2]:6.0f} max {q[3]:6.0f} us  last start {start.max():6.0f} us", flush=True)
    return t, ds

tt = torch.linspace(0, 1, 256, device=dev)
probes = o[:, None, :] + (near[:, None] + (far - near)[:, None] * tt[None, :])[..., None] * d[:, None, :]
run("probes 512 x 256 (sphere chord)", probes)
t, ds = run("probes again", probes)
ds = ds.reshape(512, 256)
hit = ds < 0.1
first = torch.where(hit.any(1), hit.float().argmax(1), torch.zeros(512, dtype=torch.long, device=dev))
last = torch.where(hit.any(1), 255 - hit.flip(1).float().argmax(1), torch.full((512,), 255, device=dev))
bn = near + (far - near) * first / 255.0
bf = near + (far - near) * last / 255.0
t64 = torch.linspace(0, 1, 64, device=dev)
coarse = o[:, None, :] + (bn[:, None] + (bf - bn)[:, None] * t64[None, :])[..., None] * d[:, None, :]
tw, _ = run("coarse 512 x 64 (bounded near/far)", coarse)
run("coarse, rays that hit only", coarse[hit.any(1)])
run("coarse, rays that miss only", coarse[~hit.any(1)])
fine = o[:, None, :] + (0.5 * (bn + bf)[:, None] + 0.02 * torch.randn(512, 16, device=dev))[..., None] * d[:, None, :]
run("16 samples per ray around mid", fine)
# slowest waves of the coarse launch: where are their queries?
dur = (tw[:, 1] - tw[:, 0]) / 100.0
worst = np.argsort(-dur)[:5]
print("slowest coarse waves (wave index -> first query's ray, sample; |x|):")
flat = coarse.reshape(-1, 3)
lanes = max(1, flat.shape[0] // max(len(dur), 1))
for w in worst:
    q0 = int(tw[w, 2]) * lanes
    print(f"  wave {int(tw[w, 2])}: {dur[w]:.0f} us, query {q0} = ray {q0 // 64} sample {q0 % 64}, |x| = {float(flat[min(q0, flat.shape[0] - 1)].norm()):.3f}, ray hits: {bool(hit.any(1)[min(q0 // 64, 511)])}")
