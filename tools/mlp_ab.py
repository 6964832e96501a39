"""tools/mlp_ab.py -- GPU box: time the MLP kernels per precision mode (nm_time_kernel, 2^20 points) and
print their field errors against the reference fixture; with --stamps also the per-phase shader-clock
durations (builds a -DNM_TESTING copy of the library under tools/_build)."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
stamps = "--stamps" in sys.argv
if stamps and "NEUMESH_HIP_LIB" not in os.environ:   # (a pre-built variant may be given through NEUMESH_HIP_LIB)
    from neumesh_amd import build as nb
    out = os.path.join(ROOT, "tools", "_build", "libneumesh_hip_stamps.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc", *nb.FLAGS, "-DNM_TESTING", os.path.join(nb.CSRC, "nm_api.hip"), "-o", out])
    os.environ["NEUMESH_HIP_LIB"] = out
import torch
import bench, common
from neumesh_amd import _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
mesh, model = bench.build_scene(140000, dev)
P = 1 << 20
rng = np.random.default_rng(0)
x = torch.from_numpy((mesh.vertices[rng.integers(0, 140000, P)] + 0.02 * rng.standard_normal((P, 3))).astype(np.float32)).to(dev)
v = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
scratch = torch.empty(int(lib.nm_field_scratch_bytes(P)), dtype=torch.uint8, device=dev)
t, keep = model.field_tables()
log = torch.zeros(32 * 16, dtype=torch.int64, device=dev)
names = {1: "geo_mlp (64 pts/wg)", 2: "geo_mlp+tangent (32 pts/wg)", 3: "colour_mlp (64 pts/wg)"}
fx = common.golden("field_v3000")
m3 = common.scene_mesh(3000)
small = common.make_model(m3, common.scene_state(m3), dev)
modes = [a for a in sys.argv[1:] if not a.startswith("--")] or ["f16x2", "f16", "fp32"]
for mode in modes:
    model.mlp_precision = small.mlp_precision = mode
    with torch.no_grad():
        q, dirs = torch.from_numpy(fx["q"]).to(dev), torch.from_numpy(fx["dirs"]).to(dev)
        sdf, nab = small.forward_with_nablas(q)
        sdf0 = small.forward_density_only(q)
        _, rgb = small.forward(q, dirs)
    print(f"[{mode}] errors vs reference fixture: sdf {np.abs(sdf.cpu().numpy() - fx['sdf']).max():.2e}, sdf(fwd-only) bit-equal to tangent kernel's: "
          f"{bool(torch.equal(sdf, sdf0))}, nabla (<= 5e-6 + 2e-4|ds| gate) {(np.abs(nab.cpu().numpy() - fx['nabla']).max(-1) - 2e-4 * np.abs(fx['ds'][:, 0])).max():.2e}, "
          f"rgb {np.abs(rgb.cpu().numpy() - fx['rgb']).max():.2e}, overflow-free: {small.check_fp16_range(force=True)}")
    for which in (1, 2, 3):
        if stamps:
            log.zero_()
            lib.nm_debug_phase_log(_lib.ptr(log))
        ms = C.c_float()
        _lib.check(lib.nm_time_kernel(model.field_handle(), model.mesh_grid.grid.handle, C.byref(t), which, _lib.ptr(x), _lib.ptr(v), P,
                                      _lib.ptr(scratch), 1 if stamps else 20, C.byref(ms), _lib.current_stream(dev)), "time")
        torch.cuda.synchronize()
        flop = {1: bench.FLOP_GEO, 2: bench.FLOP_GEO + bench.FLOP_TANGENT, 3: bench.FLOP_COL}[which]
        print(f"[{mode}] {names[which]}: {ms.value:.3f} ms per 2^20 points = {P * flop / (ms.value * 1e-3) / 1e12:.1f} algorithmic TFLOP/s")
        if stamps:
            lib.nm_debug_phase_log(None)
            a = log.cpu().numpy().reshape(32, 16)
            used = [c for c in range(16) if a[:, c].min() > 0]
            used.sort(key=lambda c: float(np.median(a[:, c] - a[:, 0])))   # in time order (the input-phase stamps use slots 10-12)
            d = np.diff(a[:, used], axis=1)
            print("      stamps", used, "median cycles per phase:", np.median(d, axis=0).astype(int).tolist(), " total",
                  int(np.median(a[:, used[-1]] - a[:, used[0]])))
