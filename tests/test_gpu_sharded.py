"""GPU (-m gpu): the ray-sharded multi-process render with the REAL HIP renderer.

  * world 1 over RCCL ("nccl"): the process-group / collective code path of bench.py --gpus N;
  * world 2: over RCCL when >= 2 devices are visible, otherwise both ranks share cuda:0 and the one
    collective goes through gloo (RCCL refuses two ranks on one device) -- pixel blocks, packing,
    padding and the all-gather are the same code either way.
Every rank must end up with the full frame, bit for bit equal to the unsharded render.
"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
port, rank, world, backend = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
import numpy as np, torch, torch.distributed as dist
import common
from neumesh_amd import synthetic
from neumesh_amd.renderer import volume_render
from neumesh_amd.rays import make_rays
from neumesh_amd.sharded import render_frame_sharded, render_frames_sharded
ndev = torch.cuda.device_count()
dev = torch.device("cuda", rank % ndev)
torch.cuda.set_device(dev)
kw = dict(init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=dev, **kw)
else:
    dist.init_process_group("gloo", **kw)
mesh = common.scene_mesh(3000)
model = common.make_model(mesh, common.scene_state(mesh), dev)
H, W = 37, 53          # 1961 pixels: odd, so the shards are ragged and the gather is padded
c2w, K = synthetic.orbit_pose(11), synthetic.pinhole_intrinsics(H, W)
rkw = dict(calc_normal=True, perturb=False, detailed_output=False, rayschunk=700)
def render(ro, rd):
    with torch.no_grad():
        return volume_render(ro, rd, model, **rkw)[2]
full = render_frame_sharded(render, c2w, K, H, W, dev)
o, d = make_rays(c2w, K, H, W, dev)      # the same device-side ray set-up the shards use (nm_make_rays), whole frame
want = render(o, d)
ok = all(torch.equal(full[k], want[k]) for k in ("rgb", "depth_volume", "mask_volume", "normals_volume"))
ok = ok and tuple(full["rgb"].shape) == (H * W, 3) and bool(torch.isfinite(full["rgb"]).all())
# the pipelined sequence (frame i's all-gather waited for after frame i + 1's render has been queued) == frame by frame
cams = [(synthetic.orbit_pose(11 + 3 * i), K) for i in range(3)]
seq = list(render_frames_sharded(render, cams, H, W, dev))
ok = ok and len(seq) == 3 and all(torch.equal(seq[0][k], full[k]) for k in ("rgb", "depth_volume", "mask_volume", "normals_volume"))
for (c, _k), fr in zip(cams[1:], seq[1:]):
    one = render_frame_sharded(render, c, K, H, W, dev)
    ok = ok and all(torch.equal(fr[k], one[k]) for k in ("rgb", "depth_volume", "mask_volume", "normals_volume"))
print("RANK", rank, "of", world, backend, "OK" if ok else "MISMATCH", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


def _run(tmp_path, world, backend):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "shard_worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), str(world), backend],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0].decode())
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append("TIMEOUT\n" + p.communicate()[0].decode())
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert all("OK" in o for o in outs), "\n".join(outs)


def test_sharded_frame_world1_rccl(cuda_device, tmp_path):
    _run(tmp_path, 1, "nccl")


def test_sharded_frame_world2_equals_unsharded(cuda_device, tmp_path):
    import torch
    _run(tmp_path, 2, "nccl" if torch.cuda.device_count() >= 2 else "gloo")


def test_sharded_frame_world8_equals_unsharded(cuda_device, tmp_path):
    """The 8 ranks of the driver's scaling run, over RCCL where 8 devices are visible, else sharing cuda:0 under gloo."""
    import torch
    _run(tmp_path, 8, "nccl" if torch.cuda.device_count() >= 8 else "gloo")


@pytest.mark.parametrize("n_ranks", [2, 8])
@pytest.mark.parametrize("shard", ["frames", "frame"])
def test_bench_two_ranks_both_partitions(cuda_device, shard, n_ranks):
    """bench.py under torch.distributed.run with 2 ranks -- over RCCL on two devices, or (one GPU visible) with --backend gloo: the
    ranks share cuda:0 and the collectives are staged through the host; the partition, per-rank timing and JSON line are the same
    code either way.  --shard frames: one frame per rank and step (weak); --shard frame: one frame per step over both ranks (strong)."""
    import json
    import torch
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    backend = "nccl" if torch.cuda.device_count() >= n_ranks else "gloo"
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")},
               HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--H", "160", "--W", "160", "--V", "3000", "--cpu-rays", "0",
           "--no-extras", "--backend", backend, "--shard", shard]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == n_ranks and line["config"]["world_size"] == n_ranks and line["config"]["backend"] == backend
    assert line["scaling"] == ("weak" if shard == "frames" else "strong") and len(line["per_rank_ms_per_step"]) == n_ranks
    rays_per_step = 160 * 160 * (n_ranks if shard == "frames" else 1)
    assert abs(line["value"] - rays_per_step / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]


def test_bench_self_spawns_for_multi_gpu(cuda_device):
    """`python bench.py --gpus 2` with no torch.distributed environment must launch itself (one rank per
    GPU); needs two devices -- on a single-GPU box only the re-exec command line is checked."""
    import torch
    if torch.cuda.device_count() < 2:
        sys.path.insert(0, ROOT)
        import bench
        assert callable(bench.self_spawn)
        pytest.skip("one GPU visible: the 2-rank bench run needs two")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--H", "200",
                          "--W", "200", "--cpu-rays", "0", "--no-extras"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    assert json.loads(line)["n_gpus"] == 2
