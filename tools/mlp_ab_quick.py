"""tools/mlp_ab_quick.py -- GPU box: nm_time_kernel + phase stamps for the libraries given on the command line
(NEUMESH_HIP_LIB is set per child process); prints one line per (library, kernel)."""
import os, subprocess, sys
if os.environ.get("NM_QUICK_CHILD"):
    import ctypes as C
    import numpy as np
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import torch, bench
    from neumesh_amd import _lib
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    mesh, model = bench.build_scene(140000, dev)
    model.mlp_precision = os.environ.get("NM_QUICK_MODE", "f16x2")
    P = 1 << 20
    rng = np.random.default_rng(0)
    x = torch.from_numpy((mesh.vertices[rng.integers(0, 140000, P)] + 0.02 * rng.standard_normal((P, 3))).astype(np.float32)).to(dev)
    v = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
    scratch = torch.empty(int(lib.nm_field_scratch_bytes(P)), dtype=torch.uint8, device=dev)
    t, keep = model.field_tables()
    log = torch.zeros(32 * 16, dtype=torch.int64, device=dev)
    for which in (1, 2, 3):
        ms = C.c_float()
        _lib.check(lib.nm_time_kernel(model.field_handle(), model.mesh_grid.grid.handle, C.byref(t), which, _lib.ptr(x), _lib.ptr(v), P,
                                      _lib.ptr(scratch), 20, C.byref(ms), _lib.current_stream(dev)), "time")
        torch.cuda.synchronize()
        msg = f"{os.environ['NM_QUICK_NAME']:>12} k{which}: {ms.value:.3f} ms"
        if hasattr(lib, "nm_debug_phase_log"):
            log.zero_()
            if lib.nm_debug_phase_log(_lib.ptr(log)) == 0:
                m1 = C.c_float()
                lib.nm_time_kernel(model.field_handle(), model.mesh_grid.grid.handle, C.byref(t), which, _lib.ptr(x), _lib.ptr(v), P, _lib.ptr(scratch), 1, C.byref(m1), _lib.current_stream(dev))
                torch.cuda.synchronize()
                lib.nm_debug_phase_log(None)
                a = log.cpu().numpy().reshape(32, 16)
                used = [c for c in range(16) if a[:, c].min() > 0]
                used.sort(key=lambda c: float(np.median(a[:, c] - a[:, 0])))
                if len(used) > 1:
                    d = np.diff(a[:, used], axis=1)
                    msg += f"  phases {np.median(d, axis=0).astype(int).tolist()} total {int(np.median(a[:, used[-1]] - a[:, used[0]]))}"
        print(msg, flush=True)
else:
    for libpath in sys.argv[1:]:
        env = dict(os.environ, NM_QUICK_CHILD="1", NEUMESH_HIP_LIB=libpath, NM_QUICK_NAME=os.path.basename(libpath).replace("lib_", "").replace(".so", ""))
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True, timeout=300)
        print("".join(l + "\n" for l in r.stdout.splitlines() if " k" in l), end="")
        if r.returncode:
            print(os.path.basename(libpath), "FAILED", r.stderr[-400:])
