"""tools/bf16_error.py -- GPU box: what a "bf16 MLP" (BASELINE.json configs[1]) costs in accuracy on this path.

The field is evaluated with the torch-op restatement twice -- fp32, and with the Linear layers under
torch.autocast(bfloat16) (bf16 operands, fp32 accumulation: what a bf16 MFMA kernel would compute) -- on the points of
the field fixture and through the whole render of the fixture rays (staged renderer, wrapper model).  Prints the numbers
DESIGN.md quotes; the split-half f16 mode the product ships is listed next to them."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common
from neumesh_amd.renderer import volume_render

dev = torch.device("cuda", 0)
mesh = common.scene_mesh(3000)
model = common.make_model(mesh, common.scene_state(mesh), dev)
fx, rf = common.golden("field_v3000"), common.golden("render_v3000_dtu")
q, dirs = torch.from_numpy(fx["q"]).to(dev), torch.from_numpy(fx["dirs"]).to(dev)


class Bf16Field(torch.nn.Module):
    """NeuMesh with its two MLPs under bf16 autocast; K-NN, distance, embeddings and compositing stay fp32."""

    def __init__(self, m):
        super().__init__()
        self.m = m
        self.enable_nablas_input = m.enable_nablas_input

    def compute_distance(self, x):
        return self.m.compute_distance(x)

    def forward_s(self):
        return self.m.forward_s()

    def _cast(self, fn, *a):
        with torch.enable_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out = fn(*a)
        return tuple(o.float().detach() if torch.is_tensor(o) else o for o in out) if isinstance(out, tuple) else out.float().detach()

    def forward_density_only(self, x):
        return self._cast(lambda x: self.m._density_autograd(x, False)[0], x)

    def forward_with_nablas(self, x):
        return self._cast(lambda x: self.m._density_autograd(x.clone(), True)[:2], x)

    def forward(self, x, v, need_nablas=True, nablas_only=False, return_ds=False):
        return self._cast(lambda x, v: self.m._forward_autograd(x.clone(), v, True, False, False), x, v)


b = Bf16Field(model)
with torch.no_grad():
    sdf32 = model.forward_density_only(q)
    _, rgb32 = model.forward(q, dirs)
sdf16 = b.forward_density_only(q)
_, rgb16 = b.forward(q, dirs)
e_sdf = (sdf16 - sdf32).abs()
print(f"bf16 MLP vs fp32 on {q.shape[0]} fixture points: |sdf| error max {float(e_sdf.max()):.2e}, mean {float(e_sdf.mean()):.2e} "
      f"(x s = 200 in the sigmoid argument: {200 * float(e_sdf.max()):.2f}); |rgb| error of the field max {float((rgb16 - rgb32).abs().max()):.2e}")
ro, rd = torch.from_numpy(rf["rays_o"]).to(dev), torch.from_numpy(rf["rays_d"]).to(dev)
kw = dict(calc_normal=True, perturb=False, detailed_output=False, rayschunk=4096)
with torch.no_grad():
    img32 = volume_render(ro, rd, model, **kw)[0]
    img16 = volume_render(ro, rd, b, **kw)[0]
ref = torch.from_numpy(rf["rgb"]).to(dev)
for name, img in (("split-half f16 (product)", img32), ("bf16 MLP", img16)):
    err = (img - ref).abs().max(-1)[0]
    mse = float(((img - ref) ** 2).mean())
    print(f"rendered fixture rays vs the reference, {name}: max |rgb| error {float(err.max()):.2e}, rays beyond 1e-4: {int((err > 1e-4).sum())}/{len(err)}, "
          f"PSNR {-10 * np.log10(mse) if mse > 0 else 200:.1f} dB")
