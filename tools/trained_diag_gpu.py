"""GPU box (round 6 diagnosis): where does the product's sdf on the TRAINED checkpoint leave the reference's (render_v140k_trained.npz: sdf_all on the
reference's own sample points)?  Per point: the fused kernels' sdf in three arithmetics, the torch-op restatement fed with the product's own
neighbours / weights / ds (separates the MLP kernels from the K-NN + distance kernel), ds itself."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common
dev = torch.device("cuda", 0)
f = common.golden("render_v140k_trained")
mesh = common.scene_mesh(140000)
model = common.make_model(mesh, common.trained_state(), dev)
dn = f["rays_d"] / np.linalg.norm(f["rays_d"], axis=-1, keepdims=True)
pts = torch.from_numpy((f["rays_o"][:, None, :] + dn[:, None, :] * f["d_all"][..., None]).astype(np.float32)).to(dev)
ref = f["sdf_all"]
out = {}
with torch.no_grad():
    for prec in ("f16x2s", "f16x2", "fp32"):
        model.mlp_precision = prec
        out[prec] = model.forward_density_only(pts)[..., 0].cpu().numpy()
    model.mlp_precision = "f16x2s"
    flat = pts.reshape(-1, 3)
    ds, idx, w = model.compute_distance(flat)
    sdf_t, _, _ = model._forward_density(flat, ds, model.geometry_features, idx, w, need_nablas=False)
    out["torch_ops_on_product_ds"] = sdf_t.reshape(ref.shape).cpu().numpy()
    out["ds"] = ds.reshape(ref.shape).cpu().numpy()
for k in ("f16x2s", "f16x2", "fp32", "torch_ops_on_product_ds"):
    e = np.abs(out[k] - ref)
    i = np.unravel_index(e.argmax(), e.shape)
    print(f"{k:26s} max |sdf - reference| {e.max():.3e} at ray {i[0]} sample {i[1]}: sdf {out[k][i]:+.6f} reference {ref[i]:+.6f} ds {out['ds'][i]:+.6f} depth {f['d_all'][i]:.5f}; "
          f"points > 3e-6: {int((e > 3e-6).sum())} of {e.size}; > 1e-5: {int((e > 1e-5).sum())}")
e = np.abs(out["f16x2s"] - ref)
order = np.argsort(e.reshape(-1))[::-1][:12]
for o in order:
    i = np.unravel_index(o, e.shape)
    print(f"  ray {i[0]:4d} sample {i[1]:3d}: err {e[i]:.2e} sdf {out['f16x2s'][i]:+.5f} fp32 {out['fp32'][i]:+.5f} torch {out['torch_ops_on_product_ds'][i]:+.5f} ref {ref[i]:+.5f} ds {out['ds'][i]:+.5f}")
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "trained_diag.npz"), **out)
