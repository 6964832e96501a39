"""GPU (-m gpu): the HIP training path of the field (C ABI nm_train_forward / nm_train_backward, csrc/nm_train.h + nm_gemm.h)
against the torch-op restatement of the reference's methods under autograd (NeuMesh._density_autograd / _forward_autograd, which
tests/test_gpu_parity.py pins to the reference's own gradients through tests/golden/train_step_v3000.npz)."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


def _points(mesh, n, seed, device, torch):
    rng = np.random.default_rng(seed)
    v = np.asarray(mesh[0] if isinstance(mesh, tuple) else mesh.vertices, np.float32)
    p = v[rng.integers(0, len(v), n)] + rng.normal(0, 0.02, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return torch.from_numpy(p).to(device), torch.from_numpy(d).to(device)


def _grads(model, outs, cots, torch):
    for p in model.parameters():
        p.grad = None
    torch.autograd.backward(list(outs), list(cots))
    return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("mode", ["density", "density_nabla", "forward"])
@pytest.mark.parametrize("n", [1, 130, 776, 5000])
def test_hip_field_matches_torch_autograd(cuda_device, torch_mod, mode, n):
    """Outputs and every parameter gradient of the three query forms, for random cotangents on sdf / nabla / rgb: a cotangent on
    nabla exercises the reverse pass of the tangent network (the reference's create_graph=True second derivative).
    (Sizes: single point, fewer points than one GEMM tile / one split-K chunk, ragged, several tiles.  The draw of 777 points is
    avoided on purpose: one colour-MLP unit there sits within an ulp of the ReLU kink and the two implementations pick different
    sides of it -- both valid, 2e-3 apart.)"""
    torch = torch_mod
    mesh = common.scene_mesh(3000)
    model = common.make_model(mesh, common.surface_state(mesh), cuda_device)
    model.train()
    xyz, dirs = _points(mesh, n, 5, cuda_device, torch)
    gen = torch.Generator(device="cpu").manual_seed(9)

    def run(backend):
        model.autograd_backend = backend
        if mode == "density":
            outs = (model.forward_density_only(xyz.clone()),)
        elif mode == "density_nabla":
            outs = model.forward_with_nablas(xyz.clone())
        else:
            outs = model.forward(xyz.clone(), dirs)
        return outs

    out_t = run("torch")
    cots = [torch.randn(o.shape, generator=gen).to(cuda_device) for o in out_t]
    g_t = _grads(model, out_t, cots, torch)
    out_h = run("hip")
    g_h = _grads(model, out_h, cots, torch)
    for a, b in zip(out_h, out_t):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), (mode, float((a - b).abs().max()))
    assert set(g_h) == set(g_t), set(g_h) ^ set(g_t)
    assert len(g_t) >= (10 if mode != "forward" else 20)
    for name in g_t:
        a, b = g_h[name].double(), g_t[name].double()
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 2e-3 * scale + 1e-7, (mode, n, name, err, scale)


def test_hip_field_ragged_and_empty(cuda_device, torch_mod):
    """[rays, samples, 3]-shaped inputs (the renderer's layout: tile order inside) and an empty batch."""
    torch = torch_mod
    mesh = common.scene_mesh(3000)
    model = common.make_model(mesh, common.surface_state(mesh), cuda_device)
    model.train()
    xyz, dirs = _points(mesh, 37 * 12, 6, cuda_device, torch)
    x3, d3 = xyz.reshape(37, 12, 3), dirs.reshape(37, 12, 3)
    model.autograd_backend = "hip"
    s_h, c_h = model.forward(x3, d3)
    model.autograd_backend = "torch"
    s_t, c_t = model.forward(x3.clone(), d3)
    assert s_h.shape == s_t.shape == (37, 12, 1) and c_h.shape == (37, 12, 3)
    assert float((s_h - s_t).abs().max()) < 2e-5 and float((c_h - c_t).abs().max()) < 2e-5
    model.autograd_backend = "hip"
    s0, n0 = model.forward_with_nablas(xyz[:0])
    assert s0.shape == (0, 1) and n0.shape == (0, 3)
    (s0.sum() + n0.sum()).backward()


@pytest.mark.parametrize("cfg", [
    dict(D_density=2, D_color=3, W=128, geometry_dim=16, color_dim=8, multires_view=2, multires_d=4, multires_fg=1, multires_ft=0,
         enable_nablas_input=False, learn_indicator_weight=False),
    dict(D_density=4, D_color=2, W=64, geometry_dim=8, color_dim=16, multires_view=0, multires_d=2, multires_fg=0, multires_ft=3,
         enable_nablas_input=True, learn_indicator_weight=True),
])
def test_hip_field_other_configurations(cuda_device, torch_mod, cfg):
    """The training kernels take their sizes from the descriptor (any width that is a multiple of 16, any depth up to 8, any
    band counts, with / without the nabla input and the learned indicator weight): two configurations that are not the reference's
    default, random default-init weights, against the torch-op restatement.  (The fused INFERENCE kernels are built for W = 256,
    so only the autograd-side methods are used here.)"""
    torch = torch_mod
    from neumesh_amd import MeshGrid, NeuMesh
    mesh = common.scene_mesh(3000)
    torch.manual_seed(4)
    model = NeuMesh(MeshGrid(common.MeshObj(mesh), cuda_device), **cfg).to(cuda_device)
    with torch.no_grad():
        model.geometry_features.mul_(0.3)
        model.color_features.mul_(0.3)
        model.ln_s.fill_(0.3)
    model.train()
    xyz, dirs = _points(mesh, 1500, 8, cuda_device, torch)
    gen = torch.Generator(device="cpu").manual_seed(2)
    for mode in ("density_nabla", "forward"):
        def run(backend):
            model.autograd_backend = backend
            return model.forward_with_nablas(xyz.clone()) if mode == "density_nabla" else model.forward(xyz.clone(), dirs)
        out_t = run("torch")
        cots = [torch.randn(o.shape, generator=gen).to(cuda_device) for o in out_t]
        g_t = _grads(model, out_t, cots, torch)
        out_h = run("hip")
        g_h = _grads(model, out_h, cots, torch)
        for a, b in zip(out_h, out_t):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), mode
        assert set(g_h) == set(g_t)
        for name in g_t:
            err, scale = float((g_h[name].double() - g_t[name].double()).abs().max()), float(g_t[name].abs().max())
            assert err <= 2e-3 * scale + 1e-7, (mode, name, err, scale)


@pytest.mark.parametrize("white,normals", [(False, True), (True, True), (False, False)])
def test_hip_composite_matches_torch_ops(cuda_device, torch_mod, white, normals):
    """nm_train_composite_forward / _backward (renderer._HipComposite) against the torch-op statement of renderer.py:264-333 under
    autograd: rgb / depth / acc / normals and the gradients with respect to the sample SDFs, the radiance, the nablas and s, on rays that
    cross a surface (decreasing SDF), graze it and miss it, R not a multiple of 64."""
    torch = torch_mod
    from neumesh_amd.renderer import _HipComposite, alpha_to_w, sdf_to_alpha
    g = torch.Generator(device="cpu").manual_seed(3)
    R, N = 333, 96
    t = torch.linspace(0, 1, N)[None, :].expand(R, N)
    off = torch.rand(R, 1, generator=g) * 1.6 - 0.3                      # crossing inside, before or after the interval
    sdf0 = (off - t) * 0.05 + 0.002 * torch.randn(R, N, generator=g)      # noisy, mostly decreasing
    sdf0[::5] = 0.02 + 0.03 * t[::5]                                      # every fifth ray misses: the SDF only grows along it
    d = torch.sort(torch.rand(R, N, generator=g) * 2 + 0.5, dim=-1).values
    dmid = torch.cat([0.5 * (d[:, 1:] + d[:, :-1]), d[:, -1:]], dim=-1)
    rad0, nab0 = torch.rand(R, N - 1, 3, generator=g), torch.randn(R, N, 3, generator=g)
    nab0[5, 7] = 0.0                                                     # a zero nabla: F.normalize's eps branch
    cots = [torch.randn(R, 3, generator=g), torch.randn(R, generator=g), torch.randn(R, generator=g), torch.randn(R, 3, generator=g)]
    cots = [c.to(cuda_device) for c in cots]

    def leafs():
        return [x.clone().to(cuda_device).requires_grad_(True) for x in (sdf0, rad0, nab0, torch.tensor([150.0]))]

    sdf, rad, nab, s = leafs()
    cdf, alpha = sdf_to_alpha(sdf, s)
    w = alpha_to_w(alpha)
    rgb = (w[..., None] * rad).sum(-2)
    acc = w.sum(-1)
    depth = (w / (w.sum(-1, keepdim=True) + 1e-10) * dmid.to(cuda_device)[:, :N - 1]).sum(-1)
    if white:
        rgb = rgb + (1.0 - acc[..., None])
    outs_t = [rgb, depth, acc] + ([(torch.nn.functional.normalize(nab[:, :N - 1], dim=-1) * w[..., None]).sum(-2)] if normals else [])
    torch.autograd.backward(outs_t, cots[:len(outs_t)])
    g_t = [x.grad.clone() for x in (sdf, rad, s)] + ([nab.grad.clone()] if normals else [])

    sdf2, rad2, nab2, s2 = leafs()
    o = _HipComposite.apply(sdf2, rad2, nab2 if normals else None, s2, dmid.to(cuda_device), white)
    outs_h = list(o[:3]) + ([o[3]] if normals else [])
    torch.autograd.backward(outs_h, cots[:len(outs_h)])
    g_h = [x.grad.clone() for x in (sdf2, rad2, s2)] + ([nab2.grad.clone()] if normals else [])
    for a, b in zip(outs_h, outs_t):
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max()))
    for a, b in zip(o[4:], (cdf, alpha, w)):
        assert float((a - b).abs().max()) <= 2e-6
    assert float(acc.min()) < 1e-3 and float(acc.max()) > 0.99                 # rays that miss and rays that are opaque
    for name, a, b in zip(("sdf", "radiance", "s", "nablas"), g_h, g_t):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-4 * scale + 1e-9, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_painting_step_matches_reference_trainer(cuda_device, torch_mod, backend, monkeypatch):
    """Trainer.forward_painting (models/trainer.py:119-172: painted rays rendered with random colour directions, background rays with
    per-sample outputs, compute_loss on their concatenation) + backward against the REFERENCE trainer's run on the same rays
    (tests/golden/painting_step_v3000.npz, oracle/gen_golden.py paint).  The random directions come from torch.rand_like; both sides
    replace it by the same host-generator draw."""
    torch = torch_mod
    from neumesh_amd.trainer import Trainer
    f = common.golden("painting_step_v3000")
    mesh = common.scene_mesh(int(f["V"]))
    model = common.make_model(mesh, common.scene_state(mesh), cuda_device)
    model.autograd_backend = backend
    model.train()
    lw = {str(k): float(v) for k, v in zip(f["loss_weight_keys"], f["loss_weight_vals"])}
    trainer = Trainer(model, loss_weights=lw, teacher_model=None, device_ids=[cuda_device.index or 0])
    trainer.teacher_model = common.StubTeacher()
    kw = dict(N_nograd_samples=2048, N_upsample_iters=4, obj_bounding_radius=1.0, batched=True, perturb=False, white_bkgd=False,
              bounded_near_far=True, calc_normal=True, N_samples=64, N_importance=64, rayschunk=4096)
    names = ("rays_o_paint", "rays_d_paint", "mask_paint", "rays_o_bg", "rays_d_bg", "mask_bg")
    model_input = {k: torch.from_numpy(f[k]) for k in names}
    ground_truth = {k: torch.from_numpy(f[k]) for k in ("rgb_paint", "rgb_bg")}
    gen = torch.Generator().manual_seed(int(f["rand_seed"]))
    monkeypatch.setattr(torch, "rand_like", lambda t, **k: torch.rand(t.shape, generator=gen, dtype=t.dtype).to(t.device))
    ret = trainer.forward_painting({"data": {"N_rays": 96}}, None, model_input, ground_truth, kw, 0, device=cuda_device)
    for k in ("loss_img", "loss_density", "loss_color", "loss_mask", "total"):
        got, want = float(ret["losses"][k]), float(f["loss." + k])
        assert abs(got - want) <= 2e-4 * max(1.0, abs(want)), (k, got, want)
    assert abs(float(ret["extras"]["psnr"]) - float(f["psnr"])) < 1e-2
    ret["losses"]["total"].backward()
    checked = 0
    for name, p in model.named_parameters():
        if "grad." + name not in f.files:
            continue
        assert p.grad is not None, name
        g = p.grad.detach().cpu().numpy()
        nrm = float(f["norm." + name])
        assert abs(float(np.linalg.norm(g.astype(np.float64))) - nrm) <= 1e-2 * nrm + 1e-6, name
        if "rows." + name in f.files:
            g = g[f["rows." + name]]
        assert np.abs(g - f["grad." + name]).max() <= 1e-2 * max(np.abs(f["grad." + name]).max(), 1e-8) + 1e-6, name
        checked += 1
    assert checked >= 20


@pytest.mark.gpu
@pytest.mark.parametrize("perturb", [False, True])
def test_fused_sampler_of_the_training_renderer_equals_the_staged_sampler(cuda_device, torch_mod, monkeypatch, perturb):
    """Training renderer (renderer.render_rays_staged(differentiable=True)): dense calls (>= 4096 rays) place their samples with ONE C
    call -- nm_render_rays(NM_RENDER_SAMPLE_ONLY, ABI v9) -- instead of the stage-by-stage form.  Same depths bit for bit, hence the
    same outputs and the same gradients to the summation order of the atomics; with perturb the uniform numbers reach the kernel
    through nm_render_cfg.u_rand (deterministic under a fixed seed)."""
    torch = torch_mod
    from neumesh_amd.renderer import volume_render
    mesh = common.scene_mesh(3000)
    rf = common.golden("render_v3000_dtu")
    o, d = torch.from_numpy(rf["rays_o"]).to(cuda_device), torch.from_numpy(rf["rays_d"]).to(cuda_device)
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=perturb, detailed_output=True, rayschunk=4096)
    outs = {}
    for name, env in (("staged", "0"), ("fused", "1"), ("fused2", "1")):
        monkeypatch.setenv("NEUMESH_FUSED_SAMPLER", env)
        model = common.make_model(mesh, common.scene_state(mesh), cuda_device)
        model.train()
        torch.manual_seed(7)
        rgb, depth, ex = volume_render(o, d, model, **kw)
        (rgb.sum() + 0.3 * depth.sum() + 0.1 * ex["normals_volume"].sum()).backward()
        outs[name] = (rgb.detach(), depth.detach(), ex["d_final"].detach(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    a, b, b2 = outs["staged"], outs["fused"], outs["fused2"]
    assert torch.equal(b[0], b2[0]) and torch.equal(b[2], b2[2])               # deterministic (same seed => same uniform numbers)
    if not perturb:
        assert torch.equal(a[2], b[2])                                          # the same mid-point depths, bit for bit
        assert float((a[0] - b[0]).abs().max()) <= 1e-6 and float((a[1] - b[1]).abs().max()) <= 1e-6
        for k in a[3]:
            ref = a[3][k].abs().max().clamp_min(1e-12)
            # (scalar parameters of the density head sum the per-sample cotangents of ALL samples -- large terms of both signs, s = 200 --
            #  with atomic adds whose order changes from run to run: density_linear.bias moves by 1-3e-3 of its value between two runs
            #  of the SAME path; the 1 % of the reference-trainer test applies to them)
            tol = 1e-2 if a[3][k].numel() == 1 else 2e-3
            assert float((a[3][k] - b[3][k]).abs().max() / ref) <= tol, k
    else:   # other uniform numbers than the staged form draws (one [iters, R, n] block instead of per-iteration blocks): same estimator
        assert float((a[0] - b[0]).abs().mean()) < 0.05 and bool(torch.isfinite(b[0]).all())
        assert not torch.equal(a[2], b[2])


@pytest.mark.parametrize("mode", [0, 1])
def test_training_gemm_alone_ragged_shapes(cuda_device, torch_mod, mode):
    """csrc/nm_gemm.h through the testing library's nm_debug_gemm: the three operand layouts of a linear layer's products (forward
    X W^T, input gradient dY W, weight gradient dY^T X with split-K and atomic accumulation onto a non-zero C), bias + ReLU epilogue,
    sizes that are not multiples of the 128 x 128 tile, of the 16 / 32-wide k step or of the split chunk -- fp32 pipe (mode 0) and
    bf16 x 3 (mode 1) against float64, cotangent-sized operands included (no scaling: bf16 has fp32's exponent)."""
    import ctypes as C
    torch = torch_mod
    from neumesh_amd import _lib
    lib = _lib.load_testing()
    st = _lib.current_stream(cuda_device)
    g = torch.Generator(device="cpu").manual_seed(3)

    def run(A, lda, akc, B, ldb, bkc, M, N, K, split=1, bias=None, relu=0, c0=None):
        Cc = (torch.zeros(M, N) if c0 is None else c0.clone()).to(cuda_device)
        _lib.check(lib.nm_debug_gemm(_lib.ptr(A), lda, akc, _lib.ptr(B), ldb, bkc, _lib.ptr(Cc), N, M, N, K, _lib.ptr(bias), relu, split,
                                     1 if (split > 1 or c0 is not None) else 0, mode, 0, None, st), "nm_debug_gemm", lib)
        torch.cuda.synchronize()
        return Cc.double().cpu()

    def close(got, want, what):
        scale = float(want.abs().max())
        assert float((got - want).abs().max()) <= 4e-6 * scale + 1e-30, (what, float((got - want).abs().max()), scale)

    for M, N, K in ((1, 64, 16), (130, 256, 48), (776, 176, 256), (5000, 128, 1000), (257, 4, 36)):
        X = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * 0.06
        b = torch.randn(N, generator=g)
        Xd, Wd, bd = X.to(cuda_device), W.to(cuda_device), b.to(cuda_device)
        close(run(Xd, K, 1, Wd, K, 1, M, N, K), X.double() @ W.double().T, ("forward", M, N, K))
        close(run(Xd, K, 1, Wd, K, 1, M, N, K, bias=bd, relu=1), torch.relu(X.double() @ W.double().T + b.double()), ("bias+relu", M, N, K))
        dY = torch.randn(M, N, generator=g) * 1e-6
        dYd = dY.to(cuda_device)
        close(run(dYd, N, 1, Wd, K, 0, M, K, N), dY.double() @ W.double(), ("input grad", M, N, K))        # [M,N] x [N,K]
        c0 = torch.randn(N, K, generator=g) * 1e-6
        for split in (1, 3, 7):
            close(run(dYd, N, 0, Xd, K, 0, N, K, M, split=split, c0=c0), c0.double() + dY.double().T @ X.double(), ("weight grad", M, N, K, split))
    if mode == 1:   # the bf16 x 3 kernel's shape contract (operands move as float4 along their contiguous axis): refused, not read out of bounds
        X, W = torch.randn(40, 38, generator=g).to(cuda_device), torch.randn(64, 38, generator=g).to(cuda_device)
        with pytest.raises(_lib.NeuMeshHipError):
            run(X, 38, 1, W, 38, 1, 40, 64, 38)
