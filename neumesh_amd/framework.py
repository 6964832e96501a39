"""Framework factory -- host-side mirror of the reference's
``models/frameworks/__init__.py:1-8`` (build_framework) and
``models/frameworks/neumesh/__init__.py:10-97`` (get_model).

``get_model(args)`` reads the same config keys, injects the same defaults with ``setdefault`` and
returns the same 5-tuple ``(model, trainer, render_kwargs_train, render_kwargs_test, renderer)``
so render.py's ``model, trainer, render_kwargs_train, render_kwargs_test, render_fn =
build_framework(args, args.model.framework)`` (render.py:272-278) and train.py's
``trainer.forward(args, indices, model_input, ground_truth, render_kwargs_train, it)`` (train.py:176)
work unchanged.  ``trainer`` is ``neumesh_amd.trainer.Trainer`` (same interface as models/trainer.py).
A NeuS teacher for the distillation losses (training.teacher_ckpt / teacher_config) is a different
framework of the reference and is built by the reference's own factory when that tree is importable.
"""
from __future__ import annotations

import copy

from .mesh_grid import MeshGrid
from .neumesh import NeuMesh
from .ply import read_ply
from .renderer import SingleRenderer
from .trainer import Trainer


def _read_mesh(path_or_mesh):
    if hasattr(path_or_mesh, "vertices"):
        return path_or_mesh
    try:  # use open3d when the caller's environment has it, exactly like the reference
        import open3d as o3d  # type: ignore
        return o3d.io.read_triangle_mesh(path_or_mesh)
    except ImportError:
        return read_ply(path_or_mesh)


def get_model(args):
    model_args = args["model"]
    mesh = _read_mesh(model_args["prior_mesh"])
    mesh_grid = MeshGrid(mesh, args["device_ids"][0], model_args.setdefault("distance_method", "frnn"))
    training, data = args["training"], args["data"]
    model_config = {
        "speed_factor": training.setdefault("speed_factor", 1.0),
        "D_density": model_args.setdefault("D_density", 3),
        "D_color": model_args.setdefault("D_color", 4),
        "W": model_args.setdefault("W", 256),
        "geometry_dim": model_args.get("geometry_dim", 32),
        "color_dim": model_args.setdefault("color_dim", 32),
        "multires_view": model_args.setdefault("multires_view", 4),
        "multires_d": model_args.setdefault("multires_d", 8),
        "multires_fg": model_args.setdefault("multires_fg", 2),
        "multires_ft": model_args.setdefault("multires_ft", 2),
        "enable_nablas_input": model_args.setdefault("enable_nablas_input", False),
        "learn_indicator_weight": model_args.get("learn_indicator_weight", False),
    }
    render_kwargs_train = {
        "N_nograd_samples": model_args.setdefault("N_nograd_samples", 2048),
        "N_upsample_iters": model_args.setdefault("N_upsample_iters", 4),
        "obj_bounding_radius": data.setdefault("obj_bounding_radius", 1.0),
        "batched": data["batch_size"] is not None,
        "perturb": model_args.setdefault("perturb", True),
        "white_bkgd": model_args.setdefault("white_bkgd", False),
        "bounded_near_far": model_args.setdefault("bounded_near_far", True),
    }
    lw = training["loss_weights"]
    for k, dflt in (("img", 0.0), ("mask", 0.0), ("eikonal", 0.0), ("distill_density", 0.0), ("distill_color", 0.0),
                    ("indicator_reg", 0.1)):
        lw.setdefault(k, dflt)
    if lw["eikonal"] > 0:  # neumesh/__init__.py:62-63: leaks into the test kwargs via the deepcopy below
        render_kwargs_train["calc_normal"] = True
    render_kwargs_test = copy.deepcopy(render_kwargs_train)
    render_kwargs_test["rayschunk"] = data["val_rayschunk"]
    render_kwargs_test["perturb"] = False
    model = NeuMesh(mesh_grid, **model_config)
    renderer = SingleRenderer(model)
    teacher_model = _load_teacher(training, model)
    trainer = Trainer(model, loss_weights=lw, teacher_model=teacher_model, device_ids=args["device_ids"])
    return model, trainer, render_kwargs_train, render_kwargs_test, renderer


def _load_teacher(training, model):
    """neumesh/__init__.py:73-89: the distillation teacher (a NeuS model of the reference) is built by the
    reference's own ``build_framework`` and its s-parameter shared with the student.  Only possible where the
    reference tree is importable (a drop-in deployment inside it); elsewhere a configured teacher is an error."""
    ckpt, cfg = training.get("teacher_ckpt"), training.get("teacher_config")
    if ckpt is None or cfg is None:
        return None
    try:
        import torch
        from models.frameworks import build_framework as ref_build   # reference tree
        from utils.io_util import load_yaml                           # reference tree
    except ImportError as e:
        raise NotImplementedError(
            "training.teacher_ckpt / teacher_config name a NeuS teacher: that framework lives in the reference tree "
            "(models/frameworks/neus), which is not importable here") from e
    teacher_config = load_yaml(cfg)
    teacher_model = ref_build(teacher_config, teacher_config.model.framework)[0]
    teacher_model.load_state_dict(torch.load(ckpt)["model"])
    model.ln_s = teacher_model.ln_s
    model.speed_factor = teacher_model.speed_factor
    return teacher_model


def build_framework(args, framework):
    if framework == "NeuMesh":
        return get_model(args)
    raise NotImplementedError(f"framework {framework!r}: only the NeuMesh render path is implemented (SURVEY.md section 8)")
