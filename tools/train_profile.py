"""tools/train_profile.py -- GPU box: where one training step (bench.py's train_step extra) spends its time (torch profiler)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from neumesh_amd import synthetic
from neumesh_amd.trainer import Trainer
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
H = W = 800
lw = {"img": 1.0, "eikonal": 0.1, "mask": 0.1, "indicator_reg": 0.1, "distill_density": 0.0, "distill_color": 0.0}
trainer = Trainer(model, loss_weights=lw, teacher_model=None, device_ids=[0])
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=float(os.environ.get("NM_TRAIN_LR", "1e-4")))
pose, K = synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W)
mi = {"intrinsics": torch.from_numpy(np.asarray(K, np.float32))[None], "c2w": torch.from_numpy(np.asarray(pose, np.float32))[None], "object_mask": torch.ones(1, H * W, dtype=torch.bool)}
gt = {"rgb": torch.full((1, H * W, 3), 0.5)}
kw = dict(N_nograd_samples=2048, N_upsample_iters=4, obj_bounding_radius=1.0, batched=True, perturb=True, white_bkgd=False, bounded_near_far=True, calc_normal=True, H=H, W=W, N_samples=64, N_importance=64, rayschunk=4096)
model.train()
def step(parts=None):
    t = [time.perf_counter()]
    opt.zero_grad(set_to_none=True)
    ret = trainer.forward({"data": {"N_rays": 512}}, None, mi, gt, kw, 0, device=dev)
    if parts is not None: torch.cuda.synchronize(); t.append(time.perf_counter())
    ret["losses"]["total"].backward()
    if parts is not None: torch.cuda.synchronize(); t.append(time.perf_counter())
    opt.step()
    if parts is not None: torch.cuda.synchronize(); t.append(time.perf_counter()); parts.append(np.diff(t))
for _ in range(3): step()
torch.cuda.synchronize()
parts = []
for _ in range(5): step(parts)
print("forward / backward / optimizer ms:", (np.mean(parts, 0) * 1e3).round(2))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(8): step()
torch.cuda.synchronize(); print("8 steps without intermediate synchronisation: ms per step", round((time.perf_counter() - t0) / 8 * 1e3, 2), flush=True)
t0 = time.perf_counter()
for _ in range(8):
    step(); torch.cuda.synchronize()
print("8 steps, one synchronisation after each step: ms per step", round((time.perf_counter() - t0) / 8 * 1e3, 2), flush=True)
if os.environ.get("NM_TRAIN_STEPS_ONLY"):
    sys.exit(0)
if os.environ.get("NM_TRAIN_AB"):
    for tag, env in (("baseline", {"NEUMESH_NO_TILE_ORDER": "1", "NEUMESH_NO_RAY_SORT": "1"}), ("tile order", {"NEUMESH_NO_RAY_SORT": "1"}), ("tile order + ray sort", {}),
                     ("baseline", {"NEUMESH_NO_TILE_ORDER": "1", "NEUMESH_NO_RAY_SORT": "1"})):
        for k in ("NEUMESH_NO_TILE_ORDER", "NEUMESH_NO_RAY_SORT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        for _ in range(2): step()
        parts = []
        for _ in range(8): step(parts)
        print(f"{tag:24s} forward / backward / optimizer ms:", (np.mean(parts, 0) * 1e3).round(2), flush=True)
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=18, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=60))
print("host-synchronising operators per 3 steps:")
for e in prof.key_averages():
    if any(k in e.key for k in ("aten::item", "aten::nonzero", "_local_scalar_dense", "aten::_to_copy", "hipMemcpy", "hipStreamSynchronize", "aten::index", "aten::masked_select")):
        print(f"  {e.key[:50]:50s} calls {e.count:5d}  cpu {e.cpu_time_total / 1e3:8.2f} ms")
