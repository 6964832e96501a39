"""Trainer -- the training-step object ``get_model`` hands back as the 2nd element of its tuple
(reference: models/frameworks/neumesh/__init__.py:91-97) and ``train.py:176`` calls as

    ret = trainer.forward(args, indices, model_input, ground_truth, render_kwargs_train, it)
    losses, extras = ret["losses"], ret["extras"]

Interface contract taken from the reference's ``models/trainer.py`` (Trainer :25-285, DensityLoss
:13-22): constructor ``Trainer(model, loss_weights, teacher_model=None, device_ids=[0], batched=True)``,
methods ``forward`` / ``forward_painting`` / ``compute_loss`` with the reference's argument lists, and the
keys of the two returned dicts (``loss_img``, ``loss_eikonal``, ``loss_density``, ``loss_color``,
``loss_indicator_vector_reg``, ``loss_mask``, ``total``; ``mask_volume_clipped``, ``psnr``,
``implicit_nablas_norm``, ``scalars``, ``select_inds``).

What runs underneath is this package's renderer: the samples of every ray are placed by the HIP stage
kernels without gradients, the field at those samples and the compositing are differentiable
(renderer.render_rays_staged), so ``ret["losses"]["total"].backward()`` reaches every parameter --
including, through the second derivative of the nabla graph, the eikonal term.
"""
from __future__ import annotations

import os
import warnings
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F

from .rays import get_rays, host_selection
from .renderer import SingleRenderer


def psnr(image_pred, image_gt, valid_mask=None, reduction="mean"):
    """-10 log10(mean squared error), as utils/metric_util.py:6-16 (reduction='none': per element)."""
    sq = (image_pred - image_gt) ** 2
    if valid_mask is not None:
        sq = sq[valid_mask]
    return -10 * torch.log10(sq.mean() if reduction == "mean" else sq)


class DensityLoss(nn.Module):
    """L1 between predicted and teacher SDF where the teacher's |SDF| <= clip (models/trainer.py:13-22)."""

    def __init__(self, density_clip=0.1):
        super().__init__()
        self.density_clip = density_clip

    def forward(self, density_pred, density_gt):
        near_surface = density_gt.abs() <= self.density_clip
        return F.l1_loss(density_gt, density_pred, reduction="none")[near_surface].mean()


def _cfg(args, group, key):
    """args.<group>.<key> for attribute-style (addict) and plain-dict configs alike."""
    g = args[group] if isinstance(args, dict) else getattr(args, group)
    return g[key] if isinstance(g, dict) else getattr(g, key)


_POOL_WARNED = False


def _warn_spinning_cpu_pool():
    """A training step is ~430 launches from one host thread.  torch's CPU back end runs large tensor operations on an OpenMP pool whose
    workers spin-wait after every parallel region; one such operation per iteration in the training process (the default collate's
    torch.stack of an image, train.py:246-260 runs the loader with num_workers=0) slows the launching thread: 16.6 -> 33-50 ms per step
    measured with 128 threads (tools/train_two_stream_stress.py).  Said once; nothing is changed on the caller's behalf."""
    global _POOL_WARNED
    if _POOL_WARNED or torch.get_num_threads() <= 16 or os.environ.get("OMP_WAIT_POLICY", "").lower() == "passive":
        return
    _POOL_WARNED = True
    warnings.warn(f"neumesh_amd.Trainer: {torch.get_num_threads()} intra-op CPU threads with an active wait policy -- large CPU tensor operations "
                  "in the training process (collate, image copies) make their workers spin beside the launch-bound training step (measured 2-3x "
                  "slower steps). Set OMP_WAIT_POLICY=passive or OMP_NUM_THREADS=8, or call torch.set_num_threads(8); see INTEGRATION.md.",
                  stacklevel=3)


class Trainer(nn.Module):
    def __init__(self, model, loss_weights, teacher_model=None, device_ids=[0], batched=True):
        super().__init__()
        self.model = model
        self.device = device_ids[0]
        renderer = SingleRenderer(model)
        # several devices: rays split along the ray dimension, as the reference does (models/trainer.py:39-42)
        self.renderer = renderer if len(device_ids) <= 1 else nn.DataParallel(renderer, device_ids=device_ids, dim=1 if batched else 0)
        self.teacher_model = teacher_model
        if teacher_model is not None:
            teacher_model.to(self.device).eval()
        self.loss_weights = loss_weights
        self.density_loss = DensityLoss()
        _warn_spinning_cpu_pool()

    # ------------------------------------------------------------------ one training step
    def forward(self, args, indices, model_input, ground_truth, render_kwargs_train: dict, it: int,
                train_progress: float = 0, device="cuda"):
        """Pick N_rays random pixels of the batch's camera(s), render them, compare with the ground truth
        (models/trainer.py:50-117)."""
        lw = self.loss_weights
        rays_o, rays_d, select_inds = get_rays(model_input["c2w"], model_input["intrinsics"], render_kwargs_train["H"], render_kwargs_train["W"],
                                               N_rays=_cfg(args, "data", "N_rays"), device=device)
        distill = lw["distill_density"] > 0 or lw["distill_color"] > 0
        rgb, _depth, extras = self.renderer(rays_o, rays_d, detailed_output=True, samples_output=distill, **render_kwargs_train)

        # Ground truth of the selected pixels: gathered WHERE THE IMAGE LIVES (the data loader hands over host tensors; the reference
        # ships the whole image to the GPU every step, trainer.py:84-91) and only the N_rays values are sent -- from pinned memory, so
        # the copy is queued instead of draining the stream.
        sel_host = None

        def pick(t, width=0):
            nonlocal sel_host
            if t.device.type == "cuda":
                idx = select_inds
            else:
                if sel_host is None:
                    sel_host = host_selection(select_inds)
                idx = sel_host
            out = torch.gather(t, 1, idx.unsqueeze(-1).expand(*idx.shape, width) if width else idx)
            return out if out.device.type == "cuda" else out.pin_memory().to(device, non_blocking=True)

        target_rgb = pick(ground_truth["rgb"], 3)
        ret = self.compute_loss(
            args, rgb, target_rgb, extras,
            mask=pick(model_input["object_mask"]) if lw["mask"] > 0 else None,
            mask_ignore=pick(model_input["mask_ignore"]) if "mask_ignore" in model_input else None,
            use_distill_loss=distill,
            use_eikonal_loss=lw["eikonal"] > 0 and "implicit_nablas" in extras,
            use_indicator_reg=lw["indicator_reg"] > 0)
        ret["extras"]["select_inds"] = select_inds
        return ret

    def forward_painting(self, args, indices, model_input, ground_truth, render_kwargs_train: dict, it: int,
                         train_progress: float = 0, device="cuda"):
        """Texture-painting fine-tune step (models/trainer.py:119-172): the painted rays are rendered with random
        colour directions, the background rays with per-sample outputs for the distillation terms."""
        def render(tag, **flags):
            ro = model_input["rays_o_" + tag].unsqueeze(1).to(device)
            rd = model_input["rays_d_" + tag].unsqueeze(1).to(device)
            rgb, _d, ex = self.renderer(ro, rd, detailed_output=True, **flags, **render_kwargs_train)
            return (rgb, ground_truth["rgb_" + tag].unsqueeze(1).to(device), model_input["mask_" + tag].unsqueeze(1).to(device), ex)

        p_rgb, p_target, p_mask, p_extras = render("paint", samples_output=False, random_color_direction=True)
        b_rgb, b_target, b_mask, b_extras = render("bg", samples_output=True, random_color_direction=False)
        b_extras["mask_volume"] = torch.cat([b_extras["mask_volume"], p_extras["mask_volume"]], dim=0)
        return self.compute_loss(args, torch.cat([p_rgb, b_rgb], dim=0), torch.cat([p_target, b_target], dim=0), b_extras,
                                 mask=torch.cat([p_mask, b_mask], dim=0), use_distill_loss=True)

    # ------------------------------------------------------------------ losses
    def _image_term(self, per_pixel, rgb, target_rgb, mask, mask_ignore):
        """Reduce the per-pixel L1 image loss and compute the PSNR over the same pixels
        (models/trainer.py:240-266)."""
        if mask is None and mask_ignore is None:
            return per_pixel.mean(), psnr(rgb, target_rgb)
        if mask is None:
            keep, kind = mask_ignore, "none"
        else:
            keep, kind = (mask if mask_ignore is None else torch.logical_and(mask, mask_ignore)), "mean"
        loss = (per_pixel * keep[..., None].float()).sum() / (keep.sum() + 1e-10)
        if kind == "mean":   # psnr(rgb[keep], target[keep]) without the host round trip of boolean indexing
            k3 = keep[..., None].float()
            mse = (((rgb - target_rgb) ** 2) * k3).sum() / (k3.sum() * rgb.shape[-1])
            return loss, -10 * torch.log10(mse)
        return loss, psnr(rgb[keep], target_rgb[keep], reduction=kind)

    def compute_loss(self, args, rgb, target_rgb, extras, mask=None, mask_ignore=None, use_eikonal_loss=False,
                     use_distill_loss=False, use_indicator_reg=False):
        """models/trainer.py:174-285.  Returns OrderedDict(losses=..., extras=...)."""
        lw = self.loss_weights
        # predicted mask close to 1 where the ground truth is 0 would explode the BCE gradient: clamp first
        acc = torch.clamp(extras["mask_volume"], 1e-3, 1 - 1e-3)
        extras["mask_volume_clipped"] = acc
        losses = OrderedDict()
        losses["loss_img"] = lw["img"] * F.l1_loss(rgb, target_rgb, reduction="none")   # reduced below
        if use_eikonal_loss:   # || d sdf / d x || = 1 at the sample points
            grad_norm = torch.norm(extras["implicit_nablas"], dim=-1)
            losses["loss_eikonal"] = lw["eikonal"] * F.mse_loss(grad_norm, torch.ones_like(grad_norm), reduction="mean")
            extras["implicit_nablas_norm"] = grad_norm
        if use_distill_loss:   # teacher field at the very sample points / directions of this render
            with torch.no_grad():
                teacher_sdf, teacher_rgb = self.teacher_model(extras["xyz"], extras["dirs"])
            losses["loss_density"] = lw["distill_density"] * F.l1_loss(extras["density"], teacher_sdf.unsqueeze(-1), reduction="mean")
            losses["loss_color"] = lw["distill_color"] * F.mse_loss(extras["colors"], teacher_rgb, reduction="mean")
        if use_indicator_reg:  # keep the learned indicator vectors near the mesh normals
            losses["loss_indicator_vector_reg"] = lw["indicator_reg"] * F.mse_loss(
                self.model.indicator_vector, self.model.mesh_grid.get_vertex_normal_torch()).mean()
        if mask is not None:
            losses["loss_mask"] = lw["mask"] * F.binary_cross_entropy(acc, mask.float(), reduction="mean")
        losses["loss_img"], extras["psnr"] = self._image_term(losses["loss_img"], rgb, target_rgb, mask, mask_ignore)
        losses["total"] = sum(losses.values())
        extras["scalars"] = {"1/s": 1.0 / self.model.forward_s().data}
        if use_indicator_reg and self.model.learn_indicator_weight:
            extras["scalars"]["indicator_weight"] = self.model.forward_indicator_weight().data
        return OrderedDict([("losses", losses), ("extras", extras)])
