"""tools/knn_wave_times_frame.py -- GPU box, -DNM_TESTING library: per-wave life of the distance kernels over ONE bench frame
(first 2^20 waves: the chained coarse pass and the first fine passes): how heavy is the tail, how long does a launch wait for it?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["NEUMESH_HIP_LIB"] = os.path.join(ROOT, "tests", "_build", "libneumesh_hip_testing.so")
import torch, bench
from neumesh_amd import synthetic, rays as R, _lib
from neumesh_amd.renderer import volume_render
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
lib = _lib.load()
H = W = 800
o, d = R.make_rays(synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W), H, W, dev)
kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=False, detailed_output=False, rayschunk=H * W)
log = torch.zeros(1 + 3 * (1 << 20), dtype=torch.int64, device=dev)
with torch.no_grad():
    volume_render(o, d, model, **kw)
    torch.cuda.synchronize()
    _lib.check(lib.nm_debug_wave_log(_lib.ptr(log)), "log")
    volume_render(o, d, model, **kw)
    torch.cuda.synchronize()
    _lib.check(lib.nm_debug_wave_log(None), "log")
n = min(int(log[0]), 1 << 20)
t = log[1:1 + 3 * n].reshape(n, 3).cpu().numpy()
order = np.argsort(t[:, 0])
t = t[order]
start, end = t[:, 0] / 100.0, t[:, 1] / 100.0
# split into launches at large gaps in start times (> 300 us without any wave starting)
cuts = [0] + list(np.nonzero(np.diff(start) > 300.0)[0] + 1) + [n]
print(f"{int(log[0])} waves logged ({n} kept), {len(cuts) - 1} launches seen")
for a, b in zip(cuts[:-1], cuts[1:]):
    if b - a < 1000:
        continue
    dur = end[a:b] - start[a:b]
    span = end[a:b].max() - start[a]
    e = np.sort(end[a:b])
    q = np.percentile(dur, [50, 90, 99, 99.9, 100])
    heavy = dur > 10 * q[0]
    print(f"launch of {b - a:7d} waves: span {span / 1e3:7.2f} ms; wave life mean {dur.mean():6.1f} p50 {q[0]:6.1f} p90 {q[1]:6.1f} p99 {q[2]:6.1f} p99.9 {q[3]:7.1f} max {q[4]:7.1f} us; "
          f"waves > 10 x median: {heavy.mean() * 100:.2f} % of waves = {dur[heavy].sum() / dur.sum() * 100:.1f} % of wave time; "
          f"last 0.1 % of the waves end {(e[-1] - e[int(0.999 * (b - a))]) :.0f} us after the rest", flush=True)
