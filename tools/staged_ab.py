import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from neumesh_amd.editing import TextureEditableNeuMesh
from neumesh_amd.renderer import make_render_cfg, render_rays_staged
dev = torch.device("cuda", 0)
mesh, model = bench.build_scene(140000, dev)
V = mesh.vertices.shape[0]
g = torch.Generator().manual_seed(5)
masks = (torch.rand(1, V, generator=g) < 0.1).to(dev)
feats = (0.1 * torch.randn(V, 32, generator=g)).to(dev)
edit = TextureEditableNeuMesh(model, [model], masks, feats)
o, d = bench.frame_rays(0, 800, 800)
ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for tag, env in (("baseline", {"NEUMESH_NO_TILE_ORDER": "1", "NEUMESH_NO_RAY_SORT": "1"}), ("tile order", {"NEUMESH_NO_RAY_SORT": "1"}), ("tile order + ray sort", {}), ("baseline", {"NEUMESH_NO_TILE_ORDER": "1", "NEUMESH_NO_RAY_SORT": "1"})):
    for k in ("NEUMESH_NO_TILE_ORDER", "NEUMESH_NO_RAY_SORT"): os.environ.pop(k, None)
    os.environ.update(env)
    with torch.no_grad():
        render_rays_staged(edit, ro[:65536], rd[:65536], make_render_cfg(calc_normal=False), 1 << 17, 1 << 20)
        torch.cuda.synchronize(); t = time.perf_counter()
        out = render_rays_staged(edit, ro, rd, make_render_cfg(calc_normal=False), 1 << 17, 1 << 20)
        torch.cuda.synchronize(); print(tag, round((time.perf_counter() - t) * 1e3, 1), "ms per staged frame", float(out["rgb"].sum()))
