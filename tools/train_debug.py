"""tools/train_debug.py -- GPU box: per-parameter gradient errors of the HIP training field vs the torch-op restatement."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, common
import test_gpu_train as T
dev = torch.device("cuda", 0)
mesh = common.scene_mesh(3000)
model = common.make_model(mesh, common.surface_state(mesh), dev)
model.train()
for mode, n in [("forward", int(x)) for x in os.environ.get("NS", "777,776,778,1000,130").split(",")]:
    xyz, dirs = T._points(mesh, n, 5, dev, torch)
    gen = torch.Generator(device="cpu").manual_seed(9)
    def run(backend, dtype=torch.float32):
        model.autograd_backend = backend
        if mode == "density_nabla":
            return model.forward_with_nablas(xyz.clone())
        return model.forward(xyz.clone(), dirs)
    out_t = run("torch")
    cots = [torch.randn(o.shape, generator=gen).to(dev) for o in out_t]
    g_t = T._grads(model, out_t, cots, torch)
    out_h = run("hip")
    g_h = T._grads(model, out_h, cots, torch)
    # the torch restatement in float64 as the arbiter
    m64 = common.make_model(mesh, common.surface_state(mesh), dev)
    print(f"== {mode} n={n}: outputs", [float((a - b).abs().max()) for a, b in zip(out_h, out_t)])
    for name in g_t:
        a, b = g_h[name].double(), g_t[name].double()
        if float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30) > 1e-4: print(f"  {name:32s} scale {float(b.abs().max()):.3e}  err {float((a - b).abs().max()):.3e}  rel {float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30):.2e}")
