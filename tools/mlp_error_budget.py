"""CPU error budget of the split-half MLP arithmetic (no GPU): the geometry / colour networks of the fixture scenes evaluated
with the operand formats and the accumulation order of nm_mlp_h2.h, against a float64 evaluation of the same weights.

Modes (per fp32 product a*b, a = activation, b = weight; every MFMA k-step = an exact 16-term dot product added to an fp32
accumulator with one rounding -- the model of v_mfma_f32_32x32x16_f16):
  f16x2   two accumulators: main += h1a*h1b ; scaled += h1a*H2b + H2a*h1b with H2 = rne16((x - h1) * 2^11); z = main + scaled * 2^-11
  f16x2s  ONE accumulator: acc += h1a*h1b + h1a*r2b + r2a*h1b with the residual halves stored UNSCALED, r2 = rne16(x - h1)
          (fp16 subnormals: quantum 2^-24; exact on the matrix pipe, tools/mfma_denorm.hip)
  f16x2c  colour network only: acc += h1a*h1b + h1a*r2b (activations rounded to 11 bits, weights 22 bits): 2 products
  f16     h1a*h1b only
  fp32    fp32 operands, fp32 sequential accumulation over k-steps (the fp32 MFMA kernels / a CPU BLAS)

Prints max / rms error of sdf (gate: 3e-6 against the reference, tests/test_gpu_parity.py) and of rgb (gate 3e-6), for the
default-init fixture weights and for the surface scene, on the points of tests/golden/field_v3000.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import common  # noqa: E402
from oracle import field as ofield  # noqa: E402

S = 100.0 / np.log(2.0)
f16, f32, f64 = np.float16, np.float32, np.float64


def split(x, scaled):
    h1 = x.astype(f16)
    r = x.astype(f64) - h1.astype(f64)
    h2 = (r * (2048.0 if scaled else 1.0)).astype(f16)
    return h1.astype(f64), h2.astype(f64)


def layer(a, W, b, mode):
    """a [P,K] fp32, W [N,K] fp32, b [N] -> z [P,N] fp32 under `mode`'s arithmetic"""
    K = a.shape[1]
    Kp = -(-K // 16) * 16
    a = np.pad(a, ((0, 0), (0, Kp - K)))
    W = np.pad(W, ((0, 0), (0, Kp - K)))
    if mode == "f64":
        return a.astype(f64) @ W.astype(f64).T + b
    if mode == "fp32":
        acc = np.broadcast_to(b.astype(f32), (a.shape[0], W.shape[0])).copy()
        for k in range(0, Kp, 16):
            acc = (acc.astype(f64) + a[:, k:k + 16].astype(f64) @ W[:, k:k + 16].astype(f64).T).astype(f32)
        return acc
    scaled = mode == "f16x2"
    a1, a2 = split(a, scaled)
    w1, w2 = split(W, scaled)
    hi = np.broadcast_to(b.astype(f32), (a.shape[0], W.shape[0])).copy()
    lo = np.zeros_like(hi)
    for k in range(0, Kp, 16):
        sl = slice(k, k + 16)
        main = a1[:, sl] @ w1[:, sl].T
        if mode == "f16":
            hi = (hi.astype(f64) + main).astype(f32)
        elif mode == "f16x2":
            hi = (hi.astype(f64) + main).astype(f32)
            lo = (lo.astype(f64) + a1[:, sl] @ w2[:, sl].T).astype(f32)       # two MFMAs into the scaled accumulator
            lo = (lo.astype(f64) + a2[:, sl] @ w1[:, sl].T).astype(f32)
        elif mode == "f16x2s":
            hi = (hi.astype(f64) + main).astype(f32)                           # three MFMAs into the one accumulator
            hi = (hi.astype(f64) + a1[:, sl] @ w2[:, sl].T).astype(f32)
            hi = (hi.astype(f64) + a2[:, sl] @ w1[:, sl].T).astype(f32)
        elif mode == "f16x2c":
            hi = (hi.astype(f64) + main).astype(f32)
            hi = (hi.astype(f64) + a1[:, sl] @ w2[:, sl].T).astype(f32)
        else:
            raise ValueError(mode)
    if mode == "f16x2":
        return (hi.astype(f64) + lo.astype(f64) / 2048.0).astype(f32)
    return hi


def softplus_l2(z):
    return np.maximum(z, np.log2(1.0 + np.exp2(np.minimum(z, 30.2965958))))


def geometry(orc, ds, idx, w, mode):
    """sdf in the kernels' log2 units: layer 0 weights and biases x S, head / S"""
    c = orc.cfg
    dt = f64 if mode == "f64" else f32
    fg = orc.interpolation(orc.state["geometry_features"], idx, w)
    h = np.concatenate([ofield.embed(ds, c.multires_d), ofield.embed(fg, c.multires_fg)], -1).astype(dt)
    for li, (W, b) in enumerate(zip(orc.geo_W, orc.geo_b)):
        Ws = (W.astype(f64) * (S if li == 0 else 1.0)).astype(f32)
        z = layer(h, Ws, (b.astype(f64) * S).astype(f32), mode)
        h = softplus_l2(z.astype(f64)).astype(dt)
    return (h.astype(f64) @ (orc.den_W.astype(f64) / S).T + orc.den_b).astype(dt)


def geometry_tangent(orc, ds, idx, w, mode, tscale):
    """(sdf, d sdf / d ds): value rows and tangent rows through the same layers; the tangent rows enter scaled by `tscale` (fp16 range)
    and are split like every other operand"""
    c = orc.cfg
    dt = f64 if mode == "f64" else f32
    fg = orc.interpolation(orc.state["geometry_features"], idx, w)
    h = np.concatenate([ofield.embed(ds, c.multires_d), ofield.embed(fg, c.multires_fg)], -1).astype(dt)
    td = [np.ones_like(ds)]
    for j in range(c.multires_d):
        fj = 2.0 ** j
        td += [fj * np.cos(ds * fj), -fj * np.sin(ds * fj)]
    t = (np.concatenate(td + [np.zeros((len(ds), h.shape[1] - len(td)))], -1) * tscale).astype(dt)
    zero = np.zeros(256, f32)
    for li, (W, b) in enumerate(zip(orc.geo_W, orc.geo_b)):
        Ws = (W.astype(f64) * (S if li == 0 else 1.0)).astype(f32)
        z = layer(h, Ws, (b.astype(f64) * S).astype(f32), mode).astype(f64)
        u = layer(t, Ws, zero, mode).astype(f64)
        e = np.exp2(np.minimum(z, 30.2965958))
        h = softplus_l2(z).astype(dt)
        t = (u * (e / (1.0 + e))).astype(dt)
    head = orc.den_W.astype(f64) / S
    return (h.astype(f64) @ head.T + orc.den_b), (t.astype(f64) @ head.T) / tscale, np.abs(t).max(), np.median(np.abs(t))


def colour(orc, ds, idx, w, nabla, dirs, mode):
    c = orc.cfg
    dt = f64 if mode == "f64" else f32
    ft = orc.interpolation(orc.state["color_features"], idx, w)
    h = np.concatenate([nabla, ofield.embed(ds, c.multires_d), ofield.embed(dirs, c.multires_view), ofield.embed(ft, c.multires_ft)], -1).astype(dt)
    for W, b in zip(orc.col_W, orc.col_b):
        h = np.maximum(layer(h, W, b, mode), 0).astype(dt)
    z = h.astype(f64) @ orc.out_W.astype(f64).T + orc.out_b
    return (1.0 / (1.0 + np.exp(-z))).astype(dt)


def main():
    fx = common.golden("field_v3000")
    mesh = common.scene_mesh(3000)
    for scene, state in (("default-init weights (noise field)", common.scene_state(mesh)), ("surface scene (sdf = ds + bump, s = 400)", common.surface_state(mesh))):
        orc = common.make_oracle(mesh, state)
        q = fx["q"]
        ds, idx, w = orc.compute_distance(q)
        near = np.abs(ds[:, 0]) < 0.25                   # the points a render visits
        _, nabla = orc.forward_with_nablas(q)
        truth_sdf = geometry(orc, ds, idx, w, "f64")
        truth_rgb = colour(orc, ds, idx, w, nabla, fx["dirs"], "f64")
        print(f"{scene}: {len(q)} points ({int(near.sum())} with |ds| < 0.25), |sdf| up to {np.abs(truth_sdf[near]).max():.3f}, s = {float(orc.forward_s()):.0f}")
        for mode in ("fp32", "f16x2", "f16x2s", "f16"):
            e = np.abs(geometry(orc, ds, idx, w, mode).astype(f64) - truth_sdf)[:, 0]
            print(f"   geometry {mode:7s}: sdf error max {e.max():.2e} (near points {e[near].max():.2e}), rms {np.sqrt((e ** 2).mean()):.2e}   [gate 3e-6]")
        _, true_t, _, _ = geometry_tangent(orc, ds, idx, w, "f64", 1.0)
        for mode, tscale in (("f16x2", 2.0 ** -15), ("f16x2s", 2.0 ** -15), ("f16x2s", 2.0 ** -11), ("f16x2s", 2.0 ** -8), ("f16x2s", 2.0 ** -6)):
            _, tt, tmax, tmed = geometry_tangent(orc, ds, idx, w, mode, tscale)
            e = np.abs(tt - true_t)[:, 0]
            print(f"   tangent  {mode:7s} scale 2^{int(np.log2(tscale))}: d sdf/d ds error max {e[near].max():.2e} (|d sdf/d ds| up to {np.abs(true_t[near]).max():.2f}); "
                  f"last layer's tangent operands: max {tmax:.2e}, median {tmed:.2e}   [nabla gate 5e-6 + 2e-4 |ds|]")
        for mode in ("fp32", "f16x2", "f16x2s", "f16x2c", "f16"):
            e = np.abs(colour(orc, ds, idx, w, nabla, fx["dirs"], mode).astype(f64) - truth_rgb).max(-1)
            print(f"   colour   {mode:7s}: rgb error max {e.max():.2e} (near points {e[near].max():.2e}), rms {np.sqrt((e ** 2).mean()):.2e}   [gate 3e-6 field, 1e-4 rendered]")
    # residual halves that fall into the fp16 subnormal range, per layer (what the unscaled single-accumulator form gives up)
    orc = common.make_oracle(mesh, common.scene_state(mesh))
    for name, Ws in (("geometry", orc.geo_W), ("colour", orc.col_W)):
        for li, W in enumerate(Ws):
            Wl = W.astype(f64) * (S if (name == "geometry" and li == 0) else 1.0)
            h1 = Wl.astype(f16).astype(f64)
            r = Wl - h1
            err_s = np.abs(r - (r * 2048).astype(f16).astype(f64) / 2048).max()
            err_u = np.abs(r - r.astype(f16).astype(f64)).max()
            print(f"   {name} layer {li}: max|w| {np.abs(Wl).max():.3f}; weight representation error: scaled residual {err_s:.1e}, unscaled residual {err_u:.1e} (2^-25 = {2.0 ** -25:.1e})")


if __name__ == "__main__":
    main()
