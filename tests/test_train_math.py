"""CPU: the DERIVATION behind the HIP training kernels (neumesh_amd/csrc/nm_train.h), restated in float64 (oracle/train_math.py), against
torch.autograd of the reference's own formulation -- nabla obtained with autograd.grad(create_graph=True) and differentiated again
(neumesh.py:204-237), the projected distance written as mesh_grid.py:125-142 writes it.  The GPU tests (tests/test_gpu_train.py) then
check that the kernels compute these formulas; this file checks that the formulas are the right ones, without a GPU."""
import numpy as np
import pytest

import common  # noqa: F401  (sys.path)

torch = pytest.importorskip("torch")
from oracle import train_math as tm  # noqa: E402


def _setup(seed=0, P=37, G=6, bands_d=3, bands_g=1, W=24, depth=3):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    x = rnd(P, 3) * 0.3
    v = x[:, None, :] + rnd(P, 8, 3) * 0.1
    n = torch.nn.functional.normalize(rnd(P, 8, 3), dim=-1).requires_grad_(True)
    w = torch.softmax(rnd(P, 8), dim=-1)
    w1 = torch.tensor(0.12, dtype=torch.float64, requires_grad=True)
    fg = (rnd(P, G) * 0.5).requires_grad_(True)
    K = (1 + 2 * bands_d) + G * (1 + 2 * bands_g)
    Ws = [(rnd(W, K if l == 0 else W) * (0.6 / np.sqrt(K if l == 0 else W))).requires_grad_(True) for l in range(depth)]
    bs = [(rnd(W) * 0.05).requires_grad_(True) for _ in range(depth)]
    wd, bd = (rnd(W) * 0.3).requires_grad_(True), (rnd(()) * 0.1).requires_grad_(True)
    return dict(x=x, v=v, n=n, w=w, w1=w1, fg=fg, Ws=Ws, bs=bs, wd=wd, bd=bd, bands_d=bands_d, bands_g=bands_g)


def _reference_sdf_nabla(s):
    """The reference's formulation: everything a function of x, nabla by autograd.grad with create_graph."""
    x = s["x"].clone().requires_grad_(True)
    d = x[:, None, :] - s["v"]
    r = d.norm(dim=-1, keepdim=True)
    middle = (s["n"] * s["w1"] + d * r) / (s["w1"] + r)                      # mesh_grid.py:137-139
    ds = (s["w"][..., None] * (d * middle).sum(-1, keepdim=True)).sum(-2)    # [P,1]
    emb_d = tm.embed(ds, s["bands_d"])[0]
    emb_g = tm.embed(s["fg"], s["bands_g"])[0]
    h = torch.cat([emb_d, emb_g], -1)
    for W, b in zip(s["Ws"], s["bs"]):
        h = tm.softplus(h @ W.T + b)
    sdf = h @ s["wd"] + s["bd"]
    nabla = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True)[0]
    return sdf, nabla


def test_pair_network_and_distance_reverse_pass_equal_second_order_autograd():
    s = _setup()
    params = [s["n"], s["w1"], s["fg"], s["wd"], s["bd"]] + s["Ws"] + s["bs"]
    g = torch.Generator().manual_seed(5)
    P = s["x"].shape[0]
    c_sdf, c_nab = torch.randn(P, generator=g, dtype=torch.float64), torch.randn(P, 3, generator=g, dtype=torch.float64)
    # reference: double backward through the nabla graph
    sdf_r, nab_r = _reference_sdf_nabla(s)
    want = torch.autograd.grad((sdf_r * c_sdf).sum() + (nab_r * c_nab).sum(), params)
    # closed form
    with torch.no_grad():
        ds, gvec = tm.distance(s["x"], s["v"], s["n"], s["w"], s["w1"])
        e_d, e_d1, e_d2 = tm.embed(ds[:, None], s["bands_d"])
        e_g, e_g1, _ = tm.embed(s["fg"], s["bands_g"])
        x0 = torch.cat([e_d, e_g], -1)
        t0 = torch.cat([e_d1, torch.zeros_like(e_g)], -1)
        sdf, alpha, saved, last = tm.geo_forward(x0, t0, s["Ws"], s["bs"], s["wd"], s["bd"])
        nabla = alpha[:, None] * gvec
        assert torch.allclose(sdf, sdf_r, rtol=0, atol=1e-12) and torch.allclose(nabla, nab_r, rtol=0, atol=1e-10)
        g_alpha = (c_nab * gvec).sum(-1)                  # nabla = alpha * g
        g_g = alpha[:, None] * c_nab
        grads, dX0, dT0 = tm.geo_backward(c_sdf, g_alpha, s["Ws"], s["wd"], saved, last)
        kd = e_d.shape[1]
        g_ds = (dX0[:, :kd] * e_d1).sum(-1) + (dT0[:, :kd] * e_d2).sum(-1)
        G = s["fg"].shape[1]
        dfg = sum(dX0[:, kd + j * G: kd + (j + 1) * G] * e_g1[:, j * G:(j + 1) * G] for j in range(1 + 2 * s["bands_g"]))
        dn, dw1 = tm.distance_backward(s["x"], s["v"], s["n"], s["w"], s["w1"], g_ds, g_g)
    got = [dn, dw1, dfg, grads["wd"], grads["bd"]] + grads["W"] + grads["b"]
    for name, a, b in zip(["indicator", "w1", "fg", "wd", "bd"] + [f"W{l}" for l in range(3)] + [f"b{l}" for l in range(3)], got, want):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 1e-9 * max(scale, 1.0), (name, float((a - b).abs().max()), scale)


def test_softplus_threshold_branch_matches_torch():
    z = torch.tensor([-1.0, -0.05, 0.0, 0.1, 0.19, 0.2000001, 0.5, 3.0], dtype=torch.float64, requires_grad=True)
    ref = torch.nn.functional.softplus(z, beta=100, threshold=20)
    assert torch.allclose(tm.softplus(z), ref, atol=1e-15)
    (g1,) = torch.autograd.grad(ref.sum(), z, create_graph=True)
    (g2,) = torch.autograd.grad(g1.sum(), z)
    s1, s2 = tm.softplus_d(z.detach())
    assert torch.allclose(s1, g1.detach(), atol=1e-12) and torch.allclose(s2, g2, atol=1e-9)


@pytest.mark.parametrize("white", [False, True])
def test_composite_reverse_scan_equals_autograd(white):
    """nm_t_composite_bwd_kernel's scan (oracle/train_math.composite_backward) against autograd of renderer.py:17-24,49-63,299-333."""
    g = torch.Generator().manual_seed(11)
    R, N = 7, 20
    t = torch.linspace(0, 1, N, dtype=torch.float64)[None, :].expand(R, N)
    sdf = ((torch.rand(R, 1, generator=g, dtype=torch.float64) * 1.4 - 0.2 - t) * 0.05 + 0.003 * torch.randn(R, N, generator=g, dtype=torch.float64))
    sdf[0] = 0.02 + 0.03 * t[0]                                   # a ray that misses
    sdf = sdf.requires_grad_(True)
    s = torch.tensor(150.0, dtype=torch.float64, requires_grad=True)
    d_mid = torch.sort(torch.rand(R, N - 1, generator=g, dtype=torch.float64) * 2 + 0.5, dim=-1).values
    rad = torch.rand(R, N - 1, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    nab = torch.randn(R, N, 3, generator=g, dtype=torch.float64).requires_grad_(True)
    cots = [torch.randn(R, 3, generator=g, dtype=torch.float64), torch.randn(R, generator=g, dtype=torch.float64),
            torch.randn(R, generator=g, dtype=torch.float64), torch.randn(R, 3, generator=g, dtype=torch.float64)]
    cdf = torch.sigmoid(sdf * s)
    alpha = ((cdf[:, :-1] - cdf[:, 1:]) / (cdf[:, :-1] + 1e-10)).clamp_min(0)
    w = alpha * torch.cumprod(torch.cat([torch.ones(R, 1, dtype=torch.float64), 1 - alpha + 1e-10], -1), -1)[:, :-1]
    rgb = (w[..., None] * rad).sum(-2)
    acc = w.sum(-1)
    depth = (w / (w.sum(-1, keepdim=True) + 1e-10) * d_mid).sum(-1)
    if white:
        rgb = rgb + (1 - acc[..., None])
    normals = (torch.nn.functional.normalize(nab[:, :N - 1], dim=-1) * w[..., None]).sum(-2)
    want = torch.autograd.grad(sum((o * c).sum() for o, c in zip((rgb, depth, acc, normals), cots)), [sdf, rad, nab, s])
    with torch.no_grad():
        got = tm.composite_backward(sdf.detach(), s.detach(), d_mid, rad.detach(), nab.detach(), white, *cots)
    assert float(acc.min()) < 1e-6 and float(acc.max()) > 0.5
    for name, a, b in zip(("sdf", "radiance", "nablas", "s"), got, want):
        assert float((a - b).abs().max()) <= 1e-9 * max(1.0, float(b.abs().max())), (name, float((a - b).abs().max()))


def test_bf16_three_piece_product_is_fp32_grade():
    """The arithmetic of the training GEMMs (csrc/nm_gemm.h, nm_gemm3_kernel), emulated in numpy: an fp32 value cut into three bf16
    pieces by truncation is reproduced EXACTLY by their sum (also values of cotangent size, 1e-9 and far below -- bf16
    shares fp32's exponent range, which is why no scale factors are needed), every piece IS a bf16 number (low 16 bits zero), and the six
    piece products the kernel adds differ from the full product by the three it drops: < 2^-21 |a||b| in the worst case of truncated
    pieces (|a2| < 2^-7 |a|, |a3| < 2^-15 |a|), 2^-23 on average.  A K = 256 dot product of such six-term products, accumulated in
    float64, is within 2^-22 of the exact one relative to sum |a||b| -- the fp32 accumulation of 256 terms on either pipe costs as
    much (tools/gemm_bench.py: 7.5e-7 of max |C| with six products, with eight, and on the fp32 pipe)."""
    rng = np.random.default_rng(0)

    def cut(a):
        a = a.astype(np.float32)
        b1 = (a.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        r1 = (a - b1).astype(np.float32)
        b2 = (r1.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        r2 = (r1 - b2).astype(np.float32)
        b3 = (r2.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        return b1, b2, b3, r2

    # (exact as long as the third piece stays a normal number: |a| >= 2^-110 ~ 8e-34)
    for scale in (1.0, 0.06, 1e-6, 1e-9, 3e-25, 1e4):
        a = (rng.standard_normal(20000) * scale).astype(np.float32)
        b = (rng.standard_normal(20000) * 0.06).astype(np.float32)
        a1, a2, a3, ar = cut(a)
        b1, b2, b3, br = cut(b)
        assert np.array_equal(ar, a3) and np.array_equal(br, b3)                      # the third piece holds all that is left
        assert np.array_equal((a1.astype(np.float64) + a2 + a3), a.astype(np.float64))  # a = a1 + a2 + a3 exactly
        for piece in (a1, a2, a3):
            assert not (piece.view(np.uint32) & np.uint32(0xFFFF)).any()
        A, B = [x.astype(np.float64) for x in (a1, a2, a3)], [x.astype(np.float64) for x in (b1, b2, b3)]
        six = A[0] * B[0] + (A[0] * B[1] + A[1] * B[0]) + (A[0] * B[2] + A[2] * B[0] + A[1] * B[1])
        full = a.astype(np.float64) * b.astype(np.float64)
        err = np.abs(six - full)
        assert (err <= 2.0 ** -21 * np.abs(full) + 1e-300).all(), scale
        assert err.mean() <= 2.0 ** -23 * np.abs(full).mean(), scale
    # a dot product of the layer width
    a = rng.standard_normal((512, 256)).astype(np.float32)
    b = (rng.standard_normal((256, 64)) * 0.06).astype(np.float32)
    A, B = cut(a)[:3], cut(b)[:3]
    A, B = [x.astype(np.float64) for x in A], [x.astype(np.float64) for x in B]
    six = A[0] @ B[0] + (A[0] @ B[1] + A[1] @ B[0]) + (A[0] @ B[2] + A[2] @ B[0] + A[1] @ B[1])
    full = a.astype(np.float64) @ b.astype(np.float64)
    mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(six - full) <= 2.0 ** -22 * mag).all()
