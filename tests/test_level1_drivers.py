"""INTEGRATION.md Level 1, EXECUTED: the reference's own drivers -- render.py's main_function / render_function (render.py:99-288) and
train.py's main_function / train (train.py:165-195, 198-460) -- run unmodified, in this process, with the ONE documented edit applied
(models/frameworks/__init__.py:1-8 returns neumesh_amd.framework.get_model's tuple), up to the first device call of the product:
there is no GPU in the build container and the product has no CPU fallback, so the run must end in NeuMeshHipError raised from INSIDE
renderer(...) / trainer.forward(...), with the reference's exact arguments on the stack.  What this proves against the live tree rather
than a recorded trace (tests/test_gpu_driver_trace.py): the config keys and setdefault mutations get_model performs are the ones the
drivers read afterwards, a checkpoint written by the REFERENCE model loads through render.py:287-288 into the product model, the
reference's get_optimizer / get_scheduler / CheckpointIO accept the product's module tree, and every keyword the drivers pass
(show_progress, detailed_output, rayschunk, H, W, N_nograd_samples ...) is accepted by the product's renderer and trainer.

Needs /root/reference (build container only); skipped elsewhere -- the GPU box has no reference tree."""
import os
import sys
import tempfile
import traceback
import types

import numpy as np
import pytest

import common
from oracle.refimport import harness

pytestmark = pytest.mark.skipif(not harness.reference_available(), reason="the reference tree exists in the build container only")

H, W = 24, 32


class _TorchProxy:
    """`torch` as the reference drivers see it on a host without a GPU: torch.device(...) is the CPU and torch.cuda.set_device a no-op
    (render.py / train.py hard-code "cuda"); everything else is torch itself."""

    class _Cuda:
        def __init__(self, cuda):
            self._c = cuda

        def set_device(self, *a, **k):
            return None

        def __getattr__(self, k):
            return getattr(self._c, k)

    def __init__(self, torch):
        self._t = torch
        self.cuda = _TorchProxy._Cuda(torch.cuda)

    def device(self, *a, **k):
        return self._t.device("cpu")

    def __getattr__(self, k):
        return getattr(self._t, k)


def _fake_dataset(torch, n=6):
    from neumesh_amd import synthetic
    K = np.eye(4, dtype=np.float32)
    K[:3, :3] = np.asarray(synthetic.pinhole_intrinsics(H, W), np.float32)[:3, :3]

    class FakeDataset(torch.utils.data.Dataset):   # what dataio.get_data returns, as far as the drivers read it
        def __init__(self):
            self.H, self.W = H, W
            self.c2w_all = [torch.from_numpy(np.asarray(synthetic.orbit_pose(7 * i), np.float32)) for i in range(n)]

        def __len__(self):
            return n

        def __getitem__(self, i):
            rng = np.random.default_rng(i)
            return i, {"intrinsics": torch.from_numpy(K.copy()), "c2w": self.c2w_all[i], "object_mask": torch.ones(H * W, dtype=torch.bool)}, \
                {"rgb": torch.from_numpy(rng.random((H * W, 3), dtype=np.float32))}
    return FakeDataset()


class _StubGridHandle:
    """Stands in for the nm_grid_t owner while the object graph is built on the CPU (the real one refuses a CPU tensor)."""
    built = []

    def __init__(self, vertices, leaf_level=0):
        self.device, self.num_vertices = vertices.device, int(vertices.shape[0])
        _StubGridHandle.built.append(self.num_vertices)

    @property
    def handle(self):
        from neumesh_amd import _lib
        raise _lib.NeuMeshHipError("no HIP device: the K-NN index lives on the GPU (no CPU fallback)")


@pytest.fixture()
def level1(monkeypatch, tmp_path):
    """The reference tree importable behind its stubs, a 3000-vertex prior mesh registered with the open3d stub, a reference checkpoint
    on disk, and the Level 1 edit applied to models/frameworks/__init__.py's build_framework."""
    path0, mods0 = list(sys.path), set(sys.modules)
    harness._activate()
    import torch
    import yaml
    import open3d as o3d_stub
    from utils.io_util import ForceKeyErrorDict
    import models.frameworks as ref_frameworks
    from neumesh_amd import framework as product_framework, mesh_grid as product_mesh_grid
    mesh = common.scene_mesh(3000)
    ref_model, _, _, _ = harness.build_reference(mesh, seed=0)
    ckpt = tmp_path / "ckpts" / "latest.pt"
    os.makedirs(ckpt.parent)
    torch.save({"model": ref_model.state_dict(), "global_step": 0, "epoch_idx": 0}, ckpt)
    with open(os.path.join(harness.REFERENCE_ROOT, "configs", "neumesh_dtu_scan63.yaml")) as f:
        cfg = yaml.safe_load(f)
    key = f"level1-mesh-{id(mesh)}"
    o3d_stub.register_mesh(key, mesh.vertices, mesh.vertex_normals)
    cfg["model"]["prior_mesh"] = key
    cfg["training"].update(teacher_ckpt=None, teacher_config=None, exp_dir=str(tmp_path), num_iters=4, i_val=2, i_backup=-1, i_save=-1,
                           ckpt_file=None, ckpt_ignore_keys=[], ckpt_only_use_keys=None, monitoring="none")
    cfg["device_ids"] = ["cpu"]
    cfg["ddp"] = False
    args = ForceKeyErrorDict(**cfg)

    def level1_build_framework(a, framework):   # INTEGRATION.md Level 1: the edited models/frameworks/__init__.py
        if framework == "NeuMesh":
            from neumesh_amd.framework import get_model
        else:
            raise NotImplementedError
        return get_model(a)

    monkeypatch.setattr(product_mesh_grid, "GridHandle", _StubGridHandle)
    monkeypatch.setattr(ref_frameworks, "build_framework", level1_build_framework)
    monkeypatch.chdir(tmp_path)
    yield types.SimpleNamespace(args=args, mesh=mesh, ref_model=ref_model, ckpt=str(ckpt), torch=torch, build=level1_build_framework,
                                product_framework=product_framework)
    # the reference tree and its stub modules (cv2, imageio, torchvision, open3d ...) must not outlive the test in this process
    stubs = os.path.join(os.path.dirname(os.path.abspath(harness.__file__)), "stubs")
    for name in set(sys.modules) - mods0:
        f = getattr(sys.modules[name], "__file__", None) or ""
        if f.startswith(harness.REFERENCE_ROOT) or f.startswith(stubs):
            del sys.modules[name]
    sys.path[:] = path0


def _frames(exc):
    return [(os.path.basename(f.filename), f.name) for f in traceback.extract_tb(exc.__traceback__)]


def test_render_py_main_function_reaches_the_products_renderer(level1, monkeypatch):
    torch = level1.torch
    import render as ref_render   # /root/reference/render.py
    from neumesh_amd import NeuMesh, SingleRenderer, _lib
    args = level1.args
    args.update(dataset_split=None, background=None, downscale=1, H=None, H_scale=None, W=None, W_scale=None, camera_path="spiral",
                test_frame=None, spiral_rad=[], num_views=2, rayschunk=4096, outbase=None, expname="level1", outdirectory=None,
                disable_rgb=False, fps=30, load_pt=level1.ckpt, device="cpu")
    seen = {}
    real_forward = SingleRenderer.forward

    def spy_forward(self, rays_o, rays_d, **kw):
        seen["kw"], seen["shape"], seen["model"] = dict(kw), tuple(rays_o.shape), self.model
        return real_forward(self, rays_o, rays_d, **kw)

    monkeypatch.setattr(SingleRenderer, "forward", spy_forward)
    monkeypatch.setattr(ref_render, "build_framework", level1.build)      # (render.py binds the name at import: from models.frameworks import build_framework)
    monkeypatch.setattr(ref_render, "get_data", lambda a, downscale=1: _fake_dataset(torch))
    monkeypatch.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)   # render.py:110,203 hard-code .cuda()
    with pytest.raises(_lib.NeuMeshHipError, match="no CPU fallback") as ei:
        ref_render.main_function(args)
    names = _frames(ei.value)
    assert ("render.py", "main_function") in names and ("render.py", "render_function") in names
    assert any(f == "renderer.py" for f, _ in names)        # raised inside the product's renderer, called by render.py:211-218
    # the reference's exact call: its rays [1, H*W, 3], its keywords
    assert seen["shape"] == (1, H * W, 3)
    kw = seen["kw"]
    assert kw["show_progress"] is True and kw["detailed_output"] is False and kw["rayschunk"] == 4096 and kw["perturb"] is False
    for k in ("N_nograd_samples", "N_upsample_iters", "obj_bounding_radius", "batched", "white_bkgd", "bounded_near_far", "calc_normal"):
        assert k in kw, k
    # render.py:287-288 loaded the REFERENCE's checkpoint into the PRODUCT's model (strict), then model.to(args.device)
    model = seen["model"]
    assert isinstance(model, NeuMesh)
    want = level1.ref_model.state_dict()
    got = model.state_dict()
    assert set(got) == set(want)
    for k in want:
        assert torch.equal(got[k], want[k]), k
    # get_model's setdefault mutations of `args` are the reference's (neumesh/__init__.py:10-97)
    ref_args = harness.build_reference(level1.mesh, seed=0)[3]
    for sec in ("model", "training", "data"):
        for k, v in ref_args[sec].items():
            if k in ("prior_mesh", "exp_dir", "teacher_ckpt", "teacher_config") or sec == "training" and k in ("num_iters", "i_val", "i_backup", "i_save", "ckpt_file", "ckpt_ignore_keys", "ckpt_only_use_keys", "monitoring"):
                continue
            assert args[sec][k] == v, (sec, k, args[sec].get(k), v)


def test_train_py_main_function_reaches_the_products_trainer(level1, monkeypatch):
    torch = level1.torch
    import train as ref_train   # /root/reference/train.py
    from utils import io_util as ref_io_util
    from neumesh_amd import _lib
    from neumesh_amd.trainer import Trainer
    args = level1.args
    args.training.ckpt_file = level1.ckpt
    args.data.batch_size = 1
    seen = {"validate": 0}
    real_fwd = Trainer.forward

    def spy_trainer_forward(self, a, indices, model_input, ground_truth, render_kwargs_train, it, **kw):
        seen.update(trainer_kw=dict(kw), it=it, render_kwargs_train=dict(render_kwargs_train), input_keys=sorted(model_input), gt_keys=sorted(ground_truth))
        return real_fwd(self, a, indices, model_input, ground_truth, render_kwargs_train, it, **kw)

    def validate_to_the_renderer(it, intrinsics, c2w, target_rgb, render_kwargs_test, volume_render_fn, logger, trainer):
        """train.py:33-57 up to its renderer call, which must end in the product's refusal; the training loop then goes on"""
        seen["validate"] += 1
        from utils import rend_util
        rays_o, rays_d, _ = rend_util.get_rays(c2w, intrinsics, render_kwargs_test["H"], render_kwargs_test["W"], N_rays=-1)
        with pytest.raises(_lib.NeuMeshHipError, match="no CPU fallback"):
            volume_render_fn(rays_o, rays_d, detailed_output=True, **render_kwargs_test)
        seen["validate_kw"] = sorted(render_kwargs_test)

    monkeypatch.setattr(ref_train, "torch", _TorchProxy(torch))
    monkeypatch.setattr(ref_train, "build_framework", level1.build)
    monkeypatch.setattr(ref_train, "get_data", lambda a, return_val=False, val_downscale=4.0: (_fake_dataset(torch), _fake_dataset(torch, 2)))
    monkeypatch.setattr(ref_train, "validate", validate_to_the_renderer)
    monkeypatch.setattr(ref_io_util, "backup", lambda d: None)             # (copies the source tree next to the experiment)
    import utils.dist_util as ref_dist_util
    monkeypatch.setattr(ref_dist_util, "torch", _TorchProxy(torch))        # init_env: torch.cuda.set_device(args.device_ids[0])
    monkeypatch.setattr(Trainer, "forward", spy_trainer_forward)
    args.device_ids = [0]            # what train.py's init_env / torch.device("cuda", local_rank) expect; the proxy maps it to the CPU
    monkeypatch.setattr(level1.product_framework, "MeshGrid", lambda mesh, device, method="frnn": __import__("neumesh_amd").mesh_grid.MeshGrid(mesh, "cpu", method))
    with pytest.raises(_lib.NeuMeshHipError, match="no CPU fallback|no HIP device") as ei:
        ref_train.main_function(args)
    names = _frames(ei.value)
    assert ("train.py", "main_function") in names and ("train.py", "train") in names
    assert any(f == "trainer.py" for f, _ in names)       # inside the product's Trainer.forward, called by train.py:176
    assert seen["validate"] == 1 and {"H", "W", "rayschunk", "perturb"} <= set(seen["validate_kw"])
    assert seen["it"] == 0 and "train_progress" in seen["trainer_kw"]
    assert seen["input_keys"] == ["c2w", "intrinsics", "object_mask"] and seen["gt_keys"] == ["rgb"]
    assert seen["render_kwargs_train"]["H"] == H and seen["render_kwargs_train"]["W"] == W
