"""tools/mlp_phases.py -- debug (GPU box): shader-clock duration of each phase of the MLP kernels."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from neumesh_amd import _lib
dev = torch.device("cuda", 0)
lib = _lib.load_testing()   # phase stamps exist in the -DNM_TESTING build only (its timed kernels write them)
mesh, model = bench.build_scene(140000, dev)
P = 1 << 20
rng = np.random.default_rng(0)
x = torch.from_numpy((mesh.vertices[rng.integers(0, 140000, P)] + 0.02 * rng.standard_normal((P, 3))).astype(np.float32)).to(dev)
v = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
scratch = torch.empty(int(lib.nm_field_scratch_bytes(P)), dtype=torch.uint8, device=dev)
t, keep = model.field_tables()
log = torch.zeros(32 * 16, dtype=torch.int64, device=dev)
names = {1: "geo_mlp (64 pts)", 2: "geo_mlp+tangent (32 pts)", 3: "colour_mlp (64 pts)"}
for which in (1, 2, 3):
    log.zero_()
    lib.nm_debug_phase_log(_lib.ptr(log))
    ms = C.c_float()
    _lib.check(lib.nm_time_kernel(model.field_handle(), model.mesh_grid.grid.handle, C.byref(t), which, _lib.ptr(x), _lib.ptr(v), P,
                                  _lib.ptr(scratch), 1, C.byref(ms), _lib.current_stream(dev)), "time")
    torch.cuda.synchronize()
    lib.nm_debug_phase_log(None)
    a = log.cpu().numpy().reshape(32, 16)
    used = [c for c in range(16) if a[:, c].min() > 0]
    d = np.diff(a[:, used], axis=1)
    print(names[which], "ms", round(ms.value, 3), "stamps", used)
    print("   median cycles per phase:", np.median(d, axis=0).astype(int).tolist(), " total", int(np.median(a[:, used[-1]] - a[:, used[0]])))
    fine = [c for c in ((0, 10, 11, 13, 14, 8, 12, 1) if which != 3 else (0, 8, 9, 10, 11, 12, 13, 1)) if a[:, c].min() > 0]   # (13 / 14 / 8: -DNM_TESTING_INPUT_STAMPS builds)
    if len(fine) > 2:
        print("   prologue stamps", fine, "median offsets from start:", np.median(a[:, fine] - a[:, [0]], axis=0).astype(int).tolist())
