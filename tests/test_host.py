"""CPU: host-side logic of the package (PLY reader, state-dict layout, factory kwargs, ray
sharding incl. a world_size-2 gloo run)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common
from neumesh_amd import ply, synthetic
from neumesh_amd.sharded import pack_outputs, shard_range, unpack_outputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ply_roundtrip_binary_and_ascii(tmp_path):
    mesh = synthetic.fibonacci_blob(200)
    tris = np.stack([np.arange(0, 198), np.arange(1, 199), np.arange(2, 200)], 1)
    p = str(tmp_path / "m.ply")
    ply.write_ply(p, mesh.vertices, tris, mesh.vertex_normals)
    m = ply.read_ply(p)
    np.testing.assert_allclose(m.vertices, mesh.vertices, atol=0)
    np.testing.assert_allclose(m.vertex_normals, mesh.vertex_normals, atol=0)
    assert np.array_equal(m.triangles, tris)
    a = str(tmp_path / "a.ply")
    with open(a, "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment x\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
                "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    m = ply.read_ply(a).compute_vertex_normals()
    assert m.triangles.shape == (2, 3)
    np.testing.assert_allclose(m.vertex_normals, np.tile([[0, 0, 1.0]], (4, 1)), atol=1e-12)


def test_state_dict_layout_matches_reference_checkpoints():
    torch = pytest.importorskip("torch")
    from neumesh_amd.neumesh import NeuMesh
    mesh = common.scene_mesh(3000)

    class FakeGrid:
        def get_number_of_vertices(self):
            return 3000

        def get_vertex_normal_torch(self):
            return torch.from_numpy(mesh.vertex_normals)

    m = NeuMesh(FakeGrid(), **common.MODEL_CFG)
    st = common.scene_state(mesh)   # keys come from the REFERENCE's state_dict (gen_golden.py)
    res = m.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in st.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m2 = NeuMesh(FakeGrid(), **{**common.MODEL_CFG, "learn_indicator_weight": True})
    assert "indicator_weight_raw" in m2.state_dict()
    # torch-op (autograd) path == reference fixture when fed the fixture's K-NN (CPU, exact same ops)
    fx = common.golden("field_v3000")
    q = torch.from_numpy(fx["q"]).requires_grad_(True)
    idx, w = torch.from_numpy(fx["idx"].astype(np.int64)), torch.from_numpy(fx["w"])
    diff = q.unsqueeze(-2) - torch.from_numpy(mesh.vertices)[idx]
    r = torch.norm(diff, dim=-1, keepdim=True)
    mid = (m.indicator_vector[idx] * 0.1 + diff * r) / (0.1 + r)
    ds = (w.unsqueeze(-1) * (diff * mid).sum(-1, keepdim=True)).sum(-2)
    sdf, nab, demb = m._forward_density(q, ds, m.geometry_features, idx, w, need_nablas=True)
    rgb = m._forward_color(demb, torch.from_numpy(fx["dirs"]), m.color_features, idx, w, nab)
    np.testing.assert_allclose(sdf.detach().numpy(), fx["sdf"], atol=1e-6)
    np.testing.assert_allclose(rgb.detach().numpy(), fx["rgb"], atol=1e-6)


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 640000, 1920001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip():
    torch = pytest.importorskip("torch")
    ret = {"rgb": torch.rand(5, 3), "depth_volume": torch.rand(5), "mask_volume": torch.rand(5), "normals_volume": torch.rand(5, 3)}
    packed, keys = pack_outputs(ret)
    assert packed.shape == (5, 8)
    back = unpack_outputs(packed, keys)
    for k in ret:
        assert torch.equal(back[k], ret[k])


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from neumesh_amd.sharded import render_sharded
WORLD = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{sys.argv[2]}", rank=int(sys.argv[3]), world_size=WORLD)
n = 1001
o = torch.arange(n * 3, dtype=torch.float32).reshape(n, 3)
d = torch.flip(o, dims=[0])
def fake_render(ro, rd):   # deterministic per-ray function standing in for the HIP renderer
    return {"rgb": ro * 0.5 + rd, "depth_volume": ro.sum(-1), "mask_volume": rd[:, 0], "normals_volume": ro - rd}
full = render_sharded(fake_render, o, d)
want = fake_render(o, d)
ok = all(torch.equal(full[k], want[k]) for k in want)
# tile-interleaved frame sharding (render_frame_sharded without the device ray set-up): each rank "renders" the pixels of
# its own tiles, one all-gather, every rank ends with the frame in pixel order
from neumesh_amd.sharded import _frame_tables, gather_tiles, gather_tiles_async, pick_tile, tile_shard_pixels
def frame_of(pix, shift):
    return fake_render(o[pix % o.shape[0]] + pix[:, None] + shift, d[pix % o.shape[0]])
for (H, W) in ((37, 53), (64, 96), (5, 3)):
    lists, per, src = _frame_tables(H, W, WORLD, pick_tile(H, W, WORLD), torch.device("cpu"))
    mine = lists[dist.get_rank()]
    assert torch.equal(torch.sort(torch.cat(lists))[0], torch.arange(H * W))
    frame = gather_tiles(frame_of(mine, 0.0), per, src, WORLD)
    allp = torch.arange(H * W)
    want_f = frame_of(allp, 0.0)
    ok = ok and all(torch.equal(frame[k], want_f[k]) for k in want_f)
    # the pipelined form (render_frames_sharded): frame i's all-gather is waited for after frame i + 1's has been posted
    waiting, got = None, []
    for i in range(3):
        nxt = gather_tiles_async(frame_of(mine, float(i)), per, src, WORLD)
        if waiting is not None:
            got.append(waiting())
        waiting = nxt
    got.append(waiting())
    for i, fr in enumerate(got):
        want_i = frame_of(allp, float(i))
        ok = ok and all(torch.equal(fr[k], want_i[k]) for k in want_i)
print("RANK", dist.get_rank(), "OK" if ok else "MISMATCH", flush=True)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
'''


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_render_gloo(tmp_path, world):
    """render_sharded, gather_tiles and the pipelined gather_tiles_async under a `world`-rank gloo group on the CPU: 2 ranks, and the 8 ranks of
    the node the driver's scaling run uses (nothing had run with 8 ranks before the driver did)."""
    pytest.importorskip("torch")
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(port), str(r), str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("OK" in o for o in outs), outs


@pytest.mark.parametrize("world", [3, 8])
@pytest.mark.parametrize("HW", [(800, 800), (1200, 1600)])
def test_tile_shards_partition_the_baseline_frames(world, HW):
    """tile_shard_pixels / _frame_tables at BASELINE's frame sizes for 3 and 8 ranks: the ranks' pixel lists partition the frame, the padded
    row count is the largest share, the scatter table inverts the gathered row order, and the shares are balanced to within one tile row."""
    torch = pytest.importorskip("torch")
    from neumesh_amd.sharded import TILE, _frame_tables, pick_tile, tile_shard_pixels
    H, W = HW
    tile = pick_tile(H, W, world)
    assert tile == TILE
    lists, per, src = _frame_tables(H, W, world, tile, torch.device("cpu"))
    assert len(lists) == world and all(torch.equal(l, tile_shard_pixels(H, W, r, world, tile)) for r, l in enumerate(lists))
    allp = torch.cat(lists)
    assert allp.numel() == H * W and torch.equal(torch.sort(allp)[0], torch.arange(H * W))
    sizes = [int(l.numel()) for l in lists]
    assert per == max(sizes) and max(sizes) - min(sizes) <= tile * tile
    gathered = torch.full((world * per,), -1, dtype=torch.int64)       # what the all-gather delivers: rank r's rows at [r * per, r * per + n_r)
    for r, l in enumerate(lists):
        gathered[r * per: r * per + l.numel()] = l
    assert torch.equal(gathered[src], torch.arange(H * W))


def test_get_rays_matches_reference_fixture_cpu():
    """oracle restatement and the package's torch-op branch of get_rays vs the reference's output."""
    torch = pytest.importorskip("torch")
    from neumesh_amd.rays import get_rays
    from oracle import render as orender
    fx = common.golden("rays_cam")
    H, W = int(fx["H"]), int(fx["W"])
    o, d = orender.get_rays(fx["c2w"], fx["intrinsics"], H, W)
    assert np.array_equal(o, fx["rays_o"]) and np.abs(d - fx["rays_d"]).max() <= 3e-7
    ro, rd, sel = get_rays(torch.from_numpy(fx["c2w"])[None], torch.from_numpy(fx["intrinsics"])[None], H, W, N_rays=-1)
    assert tuple(ro.shape) == (1, H * W, 3) and tuple(sel.shape) == (1, H * W)
    np.testing.assert_allclose(rd[0].numpy(), fx["rays_d"], atol=3e-7)
    np.testing.assert_allclose(ro[0].numpy(), fx["rays_o"], atol=0)
    ro2, rd2, sel2 = get_rays(torch.from_numpy(fx["c2w"])[None], torch.from_numpy(fx["intrinsics"])[None], H, W, N_rays=50)
    assert tuple(rd2.shape) == (1, 50, 3)
    np.testing.assert_allclose(rd2[0].numpy(), fx["rays_d"][sel2[0].numpy()], atol=3e-7)


def test_trainer_compute_loss_matches_reference_fixture():
    """neumesh_amd.trainer.Trainer.compute_loss against the REFERENCE Trainer's values on the same tensors
    (tests/golden/trainer_compute_loss.npz, oracle/gen_golden.py): every loss term, the total and the PSNR, for
    the four mask / mask_ignore combinations of models/trainer.py:240-266.  Pure torch: runs without a GPU."""
    torch = pytest.importorskip("torch")
    from neumesh_amd.trainer import Trainer
    f = common.golden("trainer_compute_loss")

    class Grid:
        def get_vertex_normal_torch(self):
            return torch.from_numpy(f["vertex_normals"])

    class Model(torch.nn.Module):
        learn_indicator_weight = False

        def __init__(self):
            super().__init__()
            self.indicator_vector = torch.nn.Parameter(torch.from_numpy(f["indicator_vector"]))
            self.mesh_grid = Grid()

        def forward_s(self):
            return torch.tensor([float(f["s"])])

    lw = {"img": 1.0, "mask": 0.1, "eikonal": 0.1, "distill_density": 1.0, "distill_color": 1.0, "indicator_reg": 0.001}
    tr = Trainer.__new__(Trainer)
    torch.nn.Module.__init__(tr)
    tr.model, tr.loss_weights, tr.teacher_model = Model(), lw, common.StubTeacher()
    t = lambda k: torch.from_numpy(f[k])
    for vname, mk, mik in (("both", t("mask"), t("mask_ignore")), ("mask_only", t("mask"), None), ("ignore_only", None, t("mask_ignore")),
                           ("none", None, None)):
        ex = {"mask_volume": t("mask_volume").clone(), "implicit_nablas": t("implicit_nablas"), "xyz": t("xyz"), "dirs": t("dirs"),
              "density": t("density"), "colors": t("colors")}
        with torch.no_grad():
            r = tr.compute_loss({}, t("rgb"), t("target"), ex, mask=mk, mask_ignore=mik, use_eikonal_loss=True, use_distill_loss=True,
                                use_indicator_reg=True)
        want = {k.split(".", 1)[1]: f[k] for k in f.files if k.startswith(vname + ".") and not k.endswith(".psnr")}
        assert list(r["losses"].keys()) == ["loss_img", "loss_eikonal", "loss_density", "loss_color", "loss_indicator_vector_reg"] + (
            ["loss_mask"] if mk is not None else []) + ["total"]
        for k, v in want.items():
            assert abs(float(r["losses"][k]) - float(v)) <= 1e-6 * max(1.0, abs(float(v))), (vname, k)
        np.testing.assert_allclose(r["extras"]["psnr"].numpy(), f[vname + ".psnr"], rtol=1e-5, atol=1e-5)
        assert set(r["extras"]) >= {"mask_volume_clipped", "psnr", "implicit_nablas_norm", "scalars"}
        assert abs(float(r["extras"]["scalars"]["1/s"]) - 1.0 / float(f["s"])) < 1e-6


def test_get_model_returns_a_trainer_with_the_reference_signature():
    """The factory's 5-tuple carries a Trainer whose forward / forward_painting / compute_loss take the reference's
    arguments (models/trainer.py:50-60, 119-129, 174-185); no GPU: only the object graph is built around a stub grid."""
    torch = pytest.importorskip("torch")
    import inspect
    from neumesh_amd.trainer import Trainer
    sig = inspect.signature(Trainer.forward)
    assert list(sig.parameters)[1:] == ["args", "indices", "model_input", "ground_truth", "render_kwargs_train", "it", "train_progress", "device"]
    assert list(inspect.signature(Trainer.forward_painting).parameters)[1:] == list(sig.parameters)[1:]
    assert list(inspect.signature(Trainer.compute_loss).parameters)[1:] == [
        "args", "rgb", "target_rgb", "extras", "mask", "mask_ignore", "use_eikonal_loss", "use_distill_loss", "use_indicator_reg"]
    assert list(inspect.signature(Trainer.__init__).parameters)[1:] == ["model", "loss_weights", "teacher_model", "device_ids", "batched"]


def test_write_png_roundtrip(tmp_path):
    """frames.write_png: an 8-bit RGB / grey PNG that decodes back to the same bytes (zlib + filter type 0)."""
    import struct, zlib
    from neumesh_amd.frames import write_png
    rng = np.random.default_rng(0)
    for shape in ((5, 7, 3), (4, 6, 1), (3, 9)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        path = str(tmp_path / "t.png")
        write_png(path, img)
        raw = open(path, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        pos, idat, ihdr = 8, b"", None
        while pos < len(raw):
            n, tag = struct.unpack(">I", raw[pos:pos + 4])[0], raw[pos + 4:pos + 8]
            data = raw[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + data) & 0xFFFFFFFF
            if tag == b"IHDR":
                ihdr = struct.unpack(">IIBBBBB", data)
            if tag == b"IDAT":
                idat += data
            pos += 12 + n
        h, w = shape[0], shape[1]
        ch = 3 if (len(shape) == 3 and shape[2] == 3) else 1
        assert ihdr == (w, h, 8, 2 if ch == 3 else 0, 0, 0, 0)
        rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
        assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(h, w, ch), img.reshape(h, w, ch))


def test_data_parallel_replicas_never_free_the_parents_field_handle(monkeypatch):
    """nn.DataParallel replicas (models/trainer.py:39-42) are shallow copies of the module: they must start without device
    state of their own (packed-weight handle, cached scalars) and the parent's nm_field_t must be destroyed exactly once,
    by its owner object, however many replicas come and go (double free through NeuMesh.__del__)."""
    torch = pytest.importorskip("torch")
    import gc
    from neumesh_amd import neumesh as nmod
    mesh = common.scene_mesh(3000)

    class FakeGrid:
        device = torch.device("cpu")

        def get_number_of_vertices(self):
            return 3000

        def get_vertex_normal_torch(self):
            return torch.from_numpy(mesh.vertex_normals)

    destroyed = []

    class FakeLib:
        def nm_field_destroy(self, h):
            destroyed.append(h)
            return 0

    monkeypatch.setattr(nmod._lib, "load", lambda require_device=True: FakeLib())
    m = nmod.NeuMesh(FakeGrid(), **common.MODEL_CFG)
    m._field = nmod.FieldHandle("HANDLE-0", torch.device("cpu"))
    m._field_key, m._scalars_key = ("k",), ("s",)
    assert not hasattr(nmod.NeuMesh, "__del__")          # ownership lives in FieldHandle alone
    for _ in range(3):                                    # one forward's worth of replicas, garbage-collected afterwards
        reps = [m._replicate_for_data_parallel() for _ in range(2)]
        for r in reps:
            assert r._field is None and r._field_key is None and r._scalars_key is None and r._is_replica
            r._field = nmod.FieldHandle(f"REPLICA-{id(r)}", torch.device("cpu"))   # what field_handle() would build on the replica's device
        del reps, r
        gc.collect()
    assert "HANDLE-0" not in destroyed and len(destroyed) == 6     # only the replicas' own handles went
    assert m._field.h == "HANDLE-0"
    del m
    gc.collect()
    assert destroyed.count("HANDLE-0") == 1


def test_fused_chunk_policy_lower_bound_and_memory_guard(monkeypatch):
    """renderer._fused_chunk (host logic, no GPU): the caller's rayschunk is a lower bound, the library's own chunk (NEUMESH_RAYSCHUNK,
    default 327 680) is halved while the lanes' workspaces would take more than a quarter of the free device memory, never below the caller's value;
    NEUMESH_RAYSCHUNK=0 honours the caller exactly."""
    import ctypes as C
    import torch
    from neumesh_amd import _lib, renderer
    lib = _lib.load(require_device=False)
    cfg = renderer.make_render_cfg(calc_normal=True)
    assert 70e3 < int(lib.nm_render_workspace_bytes(C.byref(cfg), 1 << 16)) / (1 << 16) < 90e3     # code widths not given: records of 64 + 64 floats (two mid-point sub-passes at 65 536 rays)
    cfg.code_dims = 32 | (32 << 16)                                         # what render_rays_fused sets from the model
    per_ray = int(lib.nm_render_workspace_bytes(C.byref(cfg), 1 << 16)) / (1 << 16)
    assert 42e3 < per_ray < 50e3                                             # 46 KB per ray with two sub-passes; 40 KB with three from 98 304 rays on
    assert 38e3 < int(lib.nm_render_workspace_bytes(C.byref(cfg), 320000)) / 320000 < 41e3
    free = [int(400e9)]
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (free[0], int(288e9)))
    monkeypatch.delenv("NEUMESH_RAYSCHUNK", raising=False)
    assert renderer.DEFAULT_RAYSCHUNK == 320 * 1024                                     # 20 GB of workspace per lane: two chunks of an 800x800 frame beat one call (round 5)
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0") == 320000             # render.py's 4096: the library's chunk (two EQUAL chunks of <= 327 680 rays)
    assert renderer._fused_chunk(lib, cfg, 327681, 4096, "cuda:0") == 163841             # two chunks, balanced
    free[0] = int(110e9)                                                                # a quarter of the free memory holds two 12.7 GB lanes
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0") == 320000
    free[0] = int(80e9)                                                                 # ... and here it does not: 160 000-ray chunks
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0") == 160000
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0", held_bytes=int(24e9)) == 320000   # the pool's own workspaces are not somebody else's memory
    free[0] = int(400e9)
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0", extra_per_ray=int(400e9) // 640000) == 4096   # the call's own tensors count
    monkeypatch.setenv("NEUMESH_RAYSCHUNK", str(1 << 20))
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0") == 640000            # opt-in: whole frame in one call
    assert renderer._fused_chunk(lib, cfg, 1920000, 4096, "cuda:0") == 1 << 20          # config 4: two chunks of <= 2^20 rays (a named size is taken as named)
    assert renderer._fused_chunk(lib, cfg, 500, 4096, "cuda:0") == 500
    free[0] = int(16e9)                                                                 # a nearly full device: halve until two workspaces fit
    c = renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0")
    assert 4096 <= c < 640000 and 2 * lib.nm_render_workspace_bytes(C.byref(cfg), c) <= free[0] // 2
    free[0] = int(1e6)
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0") == 4096             # never below what the caller asked for
    monkeypatch.setenv("NEUMESH_RAYSCHUNK", "0")
    free[0] = int(400e9)
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0") == 4096
    ws = renderer._Workspace()
    ws.buf = torch.empty(1000, dtype=torch.uint8)
    ws.trim(2000)
    assert ws.buf is not None
    ws.trim(999)
    assert ws.buf is None                                                               # oversized pooled workspaces go back after the call
    monkeypatch.setenv("NEUMESH_RAYSCHUNK", "100000")
    assert renderer._fused_chunk(lib, cfg, 640000, 4096, "cuda:0") == 100000
    assert renderer._fused_chunk(lib, cfg, 640000, 300000, "cuda:0") == 300000         # the caller's larger value wins


def test_bench_line_guard_prints_the_line_when_the_main_thread_stalls():
    """bench.py's watchdog (line_guard): a row after the headline that never returns must not cost the ONE line of the contract -- after
    the budget the line is printed with the rows finished so far and the process ends with status 0."""
    import json
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "out = {'metric': 'm', 'value': 1.0}; extra = {'row_a': {'ms': 2.0}}\n"
            "emit, timer = bench.line_guard(out, extra, 0.5); timer.start()\n"
            "extra['row_b'] = {'ms': 3.0}\n"
            "time.sleep(30)\n"
            "print('NOT REACHED')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "NOT REACHED" not in r.stdout
    d = json.loads(lines[0])
    assert d["value"] == 1.0 and set(d["extra"]) == {"row_a", "row_b", "_watchdog"}
    # ... and the normal end: emit() once, a second call (the timer firing late) prints nothing
    code2 = ("import sys; sys.path.insert(0, %r); import bench\n"
             "emit, timer = bench.line_guard({'value': 2.0}, {}, 60.0)\n"
             "assert emit() and not emit()\n") % ROOT
    r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=120)
    assert r2.returncode == 0 and r2.stdout.count("{") == 1, r2.stdout + r2.stderr[-300:]


def test_training_loop_fixture_schedule_and_groups():
    """CPU side of tests/test_gpu_train_loop.py: the loop helpers restated there (models/base.py:578-676, train.py:165-195) reproduce the
    learning rates the reference's get_optimizer / get_scheduler produced (tests/golden/train_loop_v3000.npz), on a module tree with
    the reference's names; the fixture's trajectory has the shape the GPU test relies on (two zero-rate iterations, then moves)."""
    import warnings
    import torch
    import test_gpu_train_loop as tl
    f = common.golden("train_loop_v3000")

    class Tree(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ln_s = torch.nn.Parameter(torch.zeros(1))
            self.color_features = torch.nn.Parameter(torch.zeros(4, 2))
            self.views_linears = torch.nn.ModuleList([torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)])
            self.pts_linears = torch.nn.ModuleList([torch.nn.Linear(2, 2)])

    m = Tree()
    lr = {str(k): float(v) for k, v in zip(f["lr_keys"], f["lr_vals"])}
    opt = tl.grouped_adam(torch, {"default": lr["default"], "color_features": lr["color_features"], "views_linears": lr["views_linears"]}, m)
    names = {id(p): n for n, p in m.named_parameters()}
    got = [[names[id(p)] for p in g["params"]] for g in opt.param_groups]
    assert got == [["ln_s", "pts_linears.0.weight", "pts_linears.0.bias"], ["color_features"],
                   ["views_linears.0.weight", "views_linears.0.bias", "views_linears.1.weight", "views_linears.1.bias"]]
    assert np.allclose([g["lr"] for g in opt.param_groups], f["group_lr0"])
    sched = tl.warmup_cosine(torch, opt, int(f["num_iters"]), int(f["warmup_steps"]))
    for it in range(int(f["n_iters"])):
        assert np.allclose([g["lr"] for g in opt.param_groups], f["lr_used"][it], rtol=1e-12, atol=0)
        opt.step()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sched.step(it)
        assert np.allclose([g["lr"] for g in opt.param_groups], f["lr_next"][it], rtol=1e-12, atol=0)
    assert (f["lr_used"][:2] == 0).all() and (f["lr_used"][2:] > 0).all()
    assert float(f["dmax.color_features"]) > 1e-3 and float(f["dmax.ln_s"]) == 0.0
    assert [str(n) for n in f["group1.names"]] == ["color_features"] and all(str(n).startswith("views_linears.") for n in f["group2.names"])


def test_render_workspace_is_at_most_40_kib_per_ray():
    """nm_render_workspace_bytes needs no device: the default layout (mid-point records of one ray sub-range at a time, nm_render_cfg.mid_passes)
    holds a 327 680-ray chunk of the headline shape in <= 13.5 GB (was 20.6 GB = 63 KB per ray); a small call is not cut."""
    import ctypes as C
    from neumesh_amd import _lib
    from neumesh_amd.renderer import make_render_cfg
    lib = _lib.load(require_device=False)
    sizes = {}
    for q in (0, 1, 2, 4, 16):
        cfg = make_render_cfg(calc_normal=True, mid_passes=q)
        cfg.code_dims = 32 | (32 << 16)
        sizes[q] = int(lib.nm_render_workspace_bytes(C.byref(cfg), 327680))
    assert sizes[4] < sizes[0] <= 13.5e9 and sizes[0] / 327680 <= 40 * 1024
    assert sizes[1] > sizes[2] > sizes[4] >= sizes[16] and sizes[1] > 20e9
    cfg = make_render_cfg(calc_normal=True)
    cfg.code_dims = 32 | (32 << 16)
    cfg1 = make_render_cfg(calc_normal=True, mid_passes=1)
    cfg1.code_dims = 32 | (32 << 16)
    assert int(lib.nm_render_workspace_bytes(C.byref(cfg), 4096)) == int(lib.nm_render_workspace_bytes(C.byref(cfg1), 4096))   # one pass below 65 536 rays
    bad = make_render_cfg(calc_normal=True, mid_passes=17)
    assert int(lib.nm_render_workspace_bytes(C.byref(bad), 4096)) < 0


def test_trained_checkpoint_is_the_one_the_fixture_was_rendered_from():
    """tests/golden/trained_v140k.pt (tools/train_field.py; utils/checkpoints.py layout) loads strictly into the product's module tree -- the
    keys / shapes a reference `render.py --load_pt` run would find (render.py:287-288) -- and is bit for bit the file render_v140k_trained.npz and
    its 32-seed sensitivity were generated from (sha256 over every tensor); the field it holds is a trained one (s = 1000, weight-norm gains
    moved off the row norms, codes no longer N(0, 1))."""
    torch = pytest.importorskip("torch")
    from neumesh_amd.neumesh import NeuMesh
    ck = torch.load(os.path.join(common.GOLDEN, "trained_v140k.pt"), map_location="cpu")
    assert set(ck) >= {"model", "global_step"} and int(ck["global_step"]) == 20000
    st = common.trained_state()
    f, sens = common.golden("render_v140k_trained"), common.golden("render_v140k_trained_sens")
    digest = common.state_digest(st)
    assert str(f["state_sha256"]) == digest and str(sens["state_sha256"]) == digest
    assert sens["self_err"].shape == (32, 1536) and np.array_equal(sens["self_err"][0], f["self_err_1ulp"])

    class FakeGrid:
        def get_number_of_vertices(self):
            return 140000

        def get_vertex_normal_torch(self):
            return torch.zeros(140000, 3)

    m = NeuMesh(FakeGrid(), **common.MODEL_CFG)
    res = m.load_state_dict(ck["model"], strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert abs(float(m.forward_s()) - 1000.0) < 1.0 and abs(float(f["s"]) - 1000.0) < 1.0
    v, g = st["pts_linears.2.0.weight_v"], st["pts_linears.2.0.weight_g"]
    assert np.abs(g[:, 0] / np.linalg.norm(v, axis=1) - 1.0).max() > 0.05          # trained: g is no longer |v|
    untrained = common.scene_state(common.scene_mesh(140000))
    assert float(np.abs(st["color_features"] - untrained["color_features"]).mean()) > 1e-3
