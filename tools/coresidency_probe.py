"""tools/coresidency_probe.py -- what would a <= 64-register K-NN traversal kernel get UNDER the MLP kernels?  (testing library)

Two MLP workgroups per CU leave 128 registers per SIMD lane: one wave of the product's K-NN kernels (96-128 registers), or TWO waves of the
plain traversal (62 registers: search only, neighbour keys out).  Round 5 measured the one-wave case end to end (a loss).  This
probe measures the two-wave case BEFORE the split traversal / epilogue kernel pair is built: the plain traversal in the pull form
(nm_debug_knn_pull) beside back-to-back MLP launches (nm_time_kernel on another stream, from another thread), with the yield protocol holding
the traversal to `keep` waves per SIMD.  Reported: K-NN ms per pass and MLP ms per launch, alone and together.

  NEUMESH_HIP_LIB=tests/_build/libneumesh_hip_testing.so python tools/coresidency_probe.py
"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NEUMESH_HIP_LIB", os.path.join(ROOT, "tests", "_build", "libneumesh_hip_testing.so"))


def main():
    import torch
    import bench
    from neumesh_amd import _lib, synthetic
    from neumesh_amd.rays import make_rays
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.load()
    mesh, model = bench.build_scene(140_000, dev, scene="surf")
    H = W = 800
    ro, rd = make_rays(synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W), H, W, dev)
    rd = torch.nn.functional.normalize(rd, dim=-1)
    # packets = 8 x 8 pixel patches at one depth (64 consecutive points), 24 depths through the object's shell
    def patches(t):
        return t.reshape(H // 8, 8, W // 8, 8, 3).permute(0, 2, 1, 3, 4).reshape(-1, 64, 3)
    o, d = patches(ro), patches(rd)
    depths = torch.linspace(1.3, 3.1, 24, device=dev)
    pts = (o[:, None] + depths[None, :, None, None] * d[:, None]).reshape(-1, 3).contiguous()     # [patch][depth][64]
    Q = pts.shape[0]
    idx = torch.empty((Q, 8), dtype=torch.int64, device=dev)
    d2 = torch.empty((Q, 8), dtype=torch.float32, device=dev)
    grid = model.grid_for(dev).grid.handle
    field = model.field_handle()
    tables, keep_alive = model.field_tables()
    P = 1 << 21
    xyz = pts[torch.randperm(Q, device=dev)[:P]].contiguous()
    scratch = torch.empty((int(lib.nm_field_scratch_bytes(P)),), dtype=torch.uint8, device=dev)
    s_mlp, s_knn = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

    def mlp(which, iters, out):
        torch.cuda.set_device(0)
        ms = C.c_float()
        _lib.check(lib.nm_time_kernel(field, grid, C.byref(tables), which, _lib.ptr(xyz), None, P, _lib.ptr(scratch), iters, C.byref(ms), C.c_void_p(s_mlp.cuda_stream)), "nm_time_kernel")
        out.append(ms.value)

    def knn(keep, yield_on, iters):
        ms = C.c_float()
        _lib.check(lib.nm_debug_knn_pull(grid, _lib.ptr(pts), Q, _lib.ptr(idx), _lib.ptr(d2), keep, yield_on, iters, C.byref(ms), C.c_void_p(s_knn.cuda_stream)), "nm_debug_knn_pull")
        return ms.value

    print(f"{Q} K-NN points in {Q // 64} packets; MLP launches of {P} points", flush=True)
    k_alone = knn(8, 0, 4)
    ref_idx = idx.clone()
    for which, name in ((1, "geometry forward"), (2, "value + tangent")):
        out = []
        mlp(which, 20, out)
        m_alone = out[0]
        print(f"[{name}] alone: K-NN pass {k_alone:.2f} ms (full occupancy), MLP launch {m_alone:.3f} ms", flush=True)
        for keep, yield_on in ((1, 1), (2, 1), (3, 1), (8, 0)):
            if yield_on:
                _lib.check(lib.nm_debug_yield_add(1), "nm_debug_yield_add")
            out = []
            n_mlp = max(8, int(3.0 * k_alone * (8 if keep < 8 else 2) / max(keep, 1) / m_alone))    # enough MLP launches to cover the K-NN pass
            th = threading.Thread(target=mlp, args=(which, n_mlp, out))
            th.start()
            time.sleep(0.02)
            k_ms = knn(keep, yield_on, 1)
            th.join()
            if yield_on:
                _lib.check(lib.nm_debug_yield_add(-1), "nm_debug_yield_add")
            same = bool(torch.equal(idx, ref_idx))
            print(f"  keep {keep} yield {yield_on}: K-NN pass {k_ms:8.2f} ms = {k_alone / k_ms:5.2f} of its full rate; MLP launch {out[0]:.3f} ms = x{out[0] / m_alone:.2f} "
                  f"({n_mlp} launches, {'all' if n_mlp * out[0] >= k_ms else 'NOT all'} of the K-NN pass covered); neighbours {'identical' if same else 'DIFFER'}", flush=True)
    del keep_alive


if __name__ == "__main__":
    main()
