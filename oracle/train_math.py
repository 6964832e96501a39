"""oracle/train_math.py -- TEST INFRASTRUCTURE ONLY: the closed-form reverse pass of the training path, restated in float64 torch ops.

Mirrors what neumesh_amd/csrc/nm_train.h computes (file:line there), independent of any GPU, so that the DERIVATION can be checked on the CPU
against torch.autograd of the reference's own formulation (neumesh.py:204-260: nabla by autograd.grad(create_graph=True); mesh_grid.py:120-142):

  pair network        z = W h + b, u = W t, h' = softplus(z), t' = softplus'(z) u                 (nm_t_softplus_kernel, nm_train_forward)
  its reverse pass    Z = H s1 + T s2 u, U = T s1, dW = Z^T h + U^T t, db = sum Z, (Z W, U W)     (nm_t_softplus_bwd_kernel, nm_train_backward)
  embedding           emb(x) = [x, sin(2^j x), cos(2^j x)], its derivative and second derivative    (nm_t_embed_kernel, nm_t_ds_emb_bwd, nm_t_code_bwd)
  projected distance  ds = sum_k w_k (w1 a_k + r_k^3) / (w1 + r_k), g = d ds / d x and their derivatives with respect to the
                      indicator vectors and w1                                                      (nm_t_distance_bwd_kernel)
"""
import torch

BETA, THRESH = 100.0, 20.0


def softplus(z):
    bz = BETA * z
    return torch.where(bz > THRESH, z, torch.log1p(torch.exp(torch.clamp(bz, max=THRESH))) / BETA)


def softplus_d(z):
    """(softplus', softplus'') as nm_t_softplus_d."""
    bz = BETA * z
    e = torch.exp(torch.clamp(bz, max=THRESH))
    s1 = torch.where(bz > THRESH, torch.ones_like(z), e / (e + 1))
    s2 = torch.where(bz > THRESH, torch.zeros_like(z), BETA * e / ((e + 1) ** 2))
    return s1, s2


def embed(x, bands):
    """[x, sin(x f0), cos(x f0), ...] over the last dimension (models/base.py:52-70), with d/dx and d2/dx2 of every output column."""
    out, d1, d2 = [x], [torch.ones_like(x)], [torch.zeros_like(x)]
    for j in range(bands):
        f = float(2 ** j)
        s, c = torch.sin(x * f), torch.cos(x * f)
        out += [s, c]
        d1 += [f * c, -f * s]
        d2 += [-f * f * s, -f * f * c]
    return torch.cat(out, -1), torch.cat(d1, -1), torch.cat(d2, -1)


def geo_forward(x0, t0, Ws, bs, wd, bd):
    """Pair network on inputs x0 [P,K], tangent inputs t0 [P,K]; returns sdf [P], alpha [P] and the saved (z, u, h, t) per layer."""
    h, t, saved = x0, t0, []
    for W, b in zip(Ws, bs):
        z, u = h @ W.T + b, t @ W.T
        s1, _ = softplus_d(z)
        saved.append((z, u, h, t))
        h, t = softplus(z), s1 * u
    return h @ wd + bd, t @ wd, saved, (h, t)


def geo_backward(g_sdf, g_alpha, Ws, wd, saved, last):
    """Reverse pass for cotangents on (sdf, alpha): gradients of every weight / bias and the cotangents of (x0, t0)."""
    h, t = last
    H, T = g_sdf[:, None] * wd[None, :], g_alpha[:, None] * wd[None, :]
    grads = {"wd": g_sdf @ h + g_alpha @ t, "bd": g_sdf.sum()}
    dWs, dbs = [], []
    for (z, u, h_in, t_in), W in zip(reversed(saved), reversed(Ws)):
        s1, s2 = softplus_d(z)
        Z, U = H * s1 + T * s2 * u, T * s1
        dWs.append(Z.T @ h_in + U.T @ t_in)
        dbs.append(Z.sum(0))
        H, T = Z @ W, U @ W
    grads["W"], grads["b"] = dWs[::-1], dbs[::-1]
    return grads, H, T


def distance(x, v, n, w, w1):
    """ds [P] and g = d ds / d x [P,3] for neighbours v, n [P,8,3], detached weights w [P,8] (mesh_grid.py:125-142 in closed form)."""
    d = x[:, None, :] - v
    r = d.norm(dim=-1)
    a = (d * n).sum(-1)
    D = w1 + r
    ds = (w * (w1 * a + r ** 3) / D).sum(-1)
    u = d / r.clamp_min(1e-30)[..., None]
    lead = w1 * n + 3 * (r ** 2)[..., None] * u
    tail = (w1 * a + r ** 3)[..., None]
    g = (w[..., None] * (lead * D[..., None] - tail * u) / (D ** 2)[..., None]).sum(1)
    return ds, g


def distance_backward(x, v, n, w, w1, g_ds, g_g):
    """Cotangents g_ds [P] of ds and g_g [P,3] of g -> (d / d n [P,8,3], d / d w1 scalar), as nm_t_distance_bwd_kernel."""
    d = x[:, None, :] - v
    r2 = (d * d).sum(-1)
    r = r2.sqrt()
    D = w1 + r
    u = d / r.clamp_min(1e-30)[..., None]
    a = (d * n).sum(-1)
    A, B = (g_g[:, None, :] * n).sum(-1), (g_g[:, None, :] * u).sum(-1)
    tail, lead = w1 * a + r * r2, w1 * A + 3 * r2 * B
    S = g_ds[:, None]
    cn = S * w1 / D - w1 * B / D ** 2
    cg = w1 / D
    dn = w[..., None] * (cn[..., None] * d + cg[..., None] * g_g[:, None, :])
    dw1 = (w * (S * (a * r - r * r2) / D ** 2 + (A * D + lead - a * B) / D ** 2 - 2 * (lead * D - tail * B) / D ** 3)).sum()
    return dn, dw1


def composite_backward(sdf, s, d_mid, rad, nab, white, g_rgb, g_depth, g_acc, g_normals):
    """Reverse scan of renderer.py:264-333 as nm_t_composite_bwd_kernel runs it (one ray after the other here): cotangents of
    rgb [R,3], depth [R], acc [R], normals [R,3] -> (g_sdf [R,N], g_rad [R,N-1,3], g_nab [R,N,3], g_s scalar)."""
    R, N = sdf.shape
    c = torch.sigmoid(sdf * s)
    q = (c[:, :-1] - c[:, 1:]) / (c[:, :-1] + 1e-10)
    a = q.clamp_min(0)
    T = torch.cumprod(torch.cat([torch.ones(R, 1, dtype=sdf.dtype), 1 - a + 1e-10], -1), -1)[:, :-1]
    w = a * T
    A = w.sum(-1)
    depth = (w * d_mid).sum(-1) / (A + 1e-10)
    ln = nab[:, :N - 1].norm(dim=-1)
    inv = 1.0 / ln.clamp_min(1e-12)
    hat = nab[:, :N - 1] * inv[..., None]
    gA = g_acc - (g_rgb.sum(-1) if white else 0.0)
    g_sdf, g_rad, g_nab = torch.zeros_like(sdf), torch.zeros_like(rad), torch.zeros_like(nab)
    g_s = sdf.new_zeros(())
    for r in range(R):
        G, carry = 0.0, 0.0
        for i in range(N - 2, -1, -1):
            dot = (g_normals[r] * hat[r, i]).sum()
            wbar = gA[r] + g_depth[r] * (d_mid[r, i] - depth[r]) / (A[r] + 1e-10) + (g_rgb[r] * rad[r, i]).sum() + dot
            g_rad[r, i] = w[r, i] * g_rgb[r]
            k = dot if ln[r, i] > 1e-12 else 0.0
            g_nab[r, i] = w[r, i] * (g_normals[r] - k * hat[r, i]) * inv[r, i]
            abar = (wbar - G) * T[r, i]
            G = wbar * a[r, i] + G * (1 - a[r, i] + 1e-10)
            c0, c1 = c[r, i], c[r, i + 1]
            den = c0 + 1e-10
            on = q[r, i] >= 0
            to_c1 = -abar / den if on else 0.0
            to_c0 = abar * (c1 + 1e-10) / den ** 2 if on else 0.0
            k1 = (carry + to_c1) * c1 * (1 - c1)
            g_sdf[r, i + 1] = k1 * s
            g_s = g_s + k1 * sdf[r, i + 1]
            carry = to_c0
        k0 = carry * c[r, 0] * (1 - c[r, 0])
        g_sdf[r, 0] = k0 * s
        g_s = g_s + k0 * sdf[r, 0]
    return g_sdf, g_rad, g_nab, g_s
