#!/usr/bin/env python
"""bench.py -- headline benchmark of the NeuMesh volumetric-render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], shape only -- there is no DTU data / checkpoint in the
environment, SURVEY.md section 8d scene S-DTU): V = 140 000-vertex prior mesh, 32-d geometry /
colour codes, W=256 MLPs at default init, s = 200; one STEP = one 800x800 frame = 640 000 rays x
(64 coarse + 64 importance) samples with bounded near/far (256 probes/ray) and normals, i.e. the
kwargs get_model() hands render.py for configs/neumesh_dtu_scan63.yaml.  Rays are resident in HBM
before the timed region.  With N GPUs every rank renders its own frame of the orbit per step
(weak scaling: per-GPU work is fixed) and the final pixels are all-gathered over RCCL -- the only
collective of the path.

Prints ONE JSON line (rank 0): value = rays/s of the whole job; `roofline` = the dominant kernel
(measured live with HIP events on the launch stream inside the timed region) against the fp32
MFMA peak; `cpu_baseline` = the CPU oracle (numpy + kd-tree K-NN) on a bounded ray sample of the
same frame (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_GEO = 353_280          # geometry MLP forward, per point
FLOP_TANGENT = 271_360      # + forward-mode tangent (nabla)
FLOP_COL = 500_736          # colour MLP
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
PEAK_HBM_GBS = 8000.0
KNN_BYTES_PER_QUERY = 76    # 12 in + 8*4 idx + 8*4 w  

MODEL_CFG = dict(D_density=3, D_color=4, W=256, geometry_dim=32, color_dim=32, multires_view=4, multires_d=8,
                 multires_fg=2, multires_ft=2, enable_nablas_input=True, speed_factor=10.0, learn_indicator_weight=False)


class _Mesh:
    def __init__(self, m):
        self.vertices, self.vertex_normals = m.vertices.astype(np.float64), m.vertex_normals.astype(np.float64)

    def compute_vertex_normals(self):
        return self


def build_scene(V, device, seed=0, s_value=200.0):
    import torch
    from neumesh_amd import MeshGrid, NeuMesh, synthetic
    mesh = synthetic.fibonacci_blob(V)
    torch.manual_seed(seed)
    model = NeuMesh(MeshGrid(_Mesh(mesh), device), **MODEL_CFG)
    with torch.no_grad():
        model.geometry_features.copy_(torch.from_numpy(synthetic.random_codes(V, 32, 1)))
        model.color_features.copy_(torch.from_numpy(synthetic.random_codes(V, 32, 2)))
        model.indicator_vector.copy_(torch.from_numpy(synthetic.noisy_indicator(mesh.vertex_normals, 3)))
        model.ln_s.fill_(float(np.log(s_value) / MODEL_CFG["speed_factor"]))
    return mesh, model.to(device).eval()


def frame_rays(frame, H, W):
    from neumesh_amd import synthetic
    return synthetic.camera_rays(synthetic.orbit_pose(frame), synthetic.pinhole_intrinsics(H, W), H, W)


def cpu_baseline(mesh, model, H, W, n_rays, gpu_rgb_frame0, rays0=None, samples=128, white_bkgd=False, calc_normal=True):
    """Oracle (CPU restatement of the reference) on a strided sample of frame 0's rays."""
    from oracle import compare, field as ofield, knn as oknn, render as orender
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    orc = ofield.OracleField(mesh.vertices, state, ofield.FieldConfig(speed_factor=MODEL_CFG["speed_factor"]))
    from scipy.spatial import cKDTree
    tree = cKDTree(mesh.vertices.astype(np.float64))
    orc.knn_fn = lambda q, v, K: oknn.knn_kdtree(q, v, K, tree=tree)
    o, d = frame_rays(0, H, W) if rays0 is None else rays0   # the very rays the GPU rendered
    sel = np.linspace(0, H * W - 1, n_rays).astype(np.int64)
    cfg = orender.RenderConfig(calc_normal=calc_normal, white_bkgd=white_bkgd, N_samples=samples // 2, N_importance=samples // 2)
    orender.render_rays(orc, o[sel[:8]], d[sel[:8]], cfg)  # warm caches / thread pools
    t = time.perf_counter()
    out = orender.render_rays(orc, o[sel], d[sel], cfg)
    dt = time.perf_counter() - t
    res = {"value": n_rays / dt, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
           "sample": f"{n_rays} rays strided over frame 0 of the same {H}x{W}x{samples} workload, {dt:.1f} s; numpy fp32 oracle + "
                     f"scipy cKDTree candidates re-ranked with the declared fp32 arithmetic (BLAS/OpenMP threads = all cores)"}
    parity = None
    if gpu_rgb_frame0 is not None:
        g = gpu_rgb_frame0[sel]
        err = np.abs(g - out["rgb"]).max(-1)
        # yardstick: how much the reference algorithm itself moves when its input rays are nudged by
        # one ulp (the up-sampling cascade + the discontinuous K-NN field amplify rounding for a few
        # rays; see oracle/compare.py and DESIGN.md "Parity")
        nudged = orender.render_rays(orc, o[sel], np.nextafter(d[sel], np.float32(10), dtype=np.float32), cfg)
        self_err = np.abs(nudged["rgb"] - out["rgb"]).max(-1)
        parity = {"rays": int(n_rays), "psnr_db": compare.psnr(g, out["rgb"]), "max_abs_rgb": float(err.max()),
                  "median_abs_rgb": float(np.median(err)), "frac_rays_within_1e-4": float((err <= 1e-4).mean()),
                  "oracle_self_sensitivity_1ulp": {"max_abs_rgb": float(self_err.max()), "median_abs_rgb": float(np.median(self_err)),
                                                   "frac_rays_within_1e-4": float((self_err <= 1e-4).mean()),
                                                   "psnr_db": compare.psnr(nudged["rgb"], out["rgb"])}}
    return res, parity


def stress5(args):
    """BASELINE config 5, the HBM-bound case of the path: V = 1 000 000 vertices, one 256-d
    vertex feature table, kernels = K-NN + gather-interpolate only (nm_distance_interpolate), queries =
    the 4096x4096 rays of a frame, one point per ray where it meets the surface shell.  One step = one
    such frame, in slabs of 256 image rows (the 1 KiB/query output of a slab is 1 GiB)."""
    import torch
    import torch.distributed as dist
    from neumesh_amd import _lib, synthetic
    from neumesh_amd.mesh_grid import MeshGrid
    from neumesh_amd.rays import make_rays
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()
    V, dim, H, W, slab = 1_000_000, 256, 4096, 4096, 256
    mesh = synthetic.fibonacci_blob(V)
    grid = MeshGrid(_Mesh(mesh), dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    table = torch.randn((V, dim), generator=gen, device=dev)
    ind = grid.vertex_normals.contiguous()
    K = synthetic.pinhole_intrinsics(H, W)
    total = args.warmup + args.steps
    feat = torch.empty((slab * W, dim), device=dev)
    ds = torch.empty((slab * W,), device=dev)

    def frame_points(f):  # resident before timing: [H*W,3] points on the r = 0.75 shell along the rays of frame f
        o, d = make_rays(synthetic.orbit_pose(f * world + rank), K, H, W, dev)
        d = torch.nn.functional.normalize(d, dim=-1)
        b = (o * d).sum(-1)
        t = -b - torch.sqrt(torch.clamp(b * b - ((o * o).sum(-1) - 0.75 ** 2), min=0.0))
        return (o + t[:, None] * d).contiguous()

    pts = [frame_points(f) for f in range(total)]
    stream = _lib.current_stream(dev)

    def step(i):
        for r0 in range(0, H, slab):
            q = pts[i][r0 * W:(r0 + slab) * W]
            _lib.check(lib.nm_distance_interpolate(grid.grid.handle, _lib.ptr(q), q.shape[0], _lib.ptr(ind), 0.1, _lib.ptr(table), dim,
                                                   _lib.ptr(ds), None, None, _lib.ptr(feat), stream), "nm_distance_interpolate")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    lib.nm_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms, n, u = C.c_double(), C.c_int64(), C.c_int64()
    _lib.check(lib.nm_profile_read(0, C.byref(ms), C.byref(n), C.byref(u)), "nm_profile_read")
    lib.nm_profile_enable(0)
    if rank == 0:
        bytes_q = 12 + 8 * dim * 4          # query + 8 gathered rows (the 4*dim-byte output row is extra)
        per_launch_q = u.value / max(n.value, 1)
        avg_ms = ms.value / max(n.value, 1)
        achieved = per_launch_q * bytes_q / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic_stress5.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("knn_distance", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        print(json.dumps({
            "metric": "K-NN + gather-interpolate queries/sec, 1M-vertex mesh x 256-d features, 4096x4096 rays (BASELINE config 5)",
            "value": world * H * W * args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"stress5: V={V}, {dim}-d table ({V * dim * 4 / 2**30:.2f} GiB), {H}x{W} queries per step per GPU in slabs of {slab} rows",
                       "parallelism": f"queries sharded: {world} GPU(s) x 1 frame per step, no collective"},
            "roofline": {"bound": "hbm", "kernel": "nm_distance_kernel<false> (K-NN + weights + 8-row gather-interpolate)",
                         "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                         "bytes_per_query": bytes_q, "written_bytes_per_query_not_counted": dim * 4 + 4,
                         "avg_launch_ms": avg_ms, "launches": n.value, "traffic": traffic}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["frame", "stress5"], default="frame",
                    help="frame = BASELINE configs[1], the headline 800x800x128 render (default); stress5 = BASELINE configs[4], "
                         "the HBM-bound K-NN + 256-d gather stress (a second roofline, not the headline metric)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--H", type=int, default=800)
    ap.add_argument("--W", type=int, default=800)
    ap.add_argument("--V", type=int, default=140_000)
    ap.add_argument("--rayschunk", type=int, default=0,
                    help="rays per nm_render_rays call; 0 = the whole frame in one call (56 KB of workspace per ray: 36 GB for 800x800)")
    ap.add_argument("--mlp-precision", choices=["f16x2", "fp32"], default="f16x2",
                    help="MLP arithmetic: split-half f16 MFMA (default; 22-bit operands, fp32 accumulation) or fp32 MFMA")
    ap.add_argument("--cpu-rays", type=int, default=1536, help="rays of the CPU-baseline sample (0 disables)")
    ap.add_argument("--samples", type=int, default=128, help="samples per ray, half coarse / half importance (BASELINE configs[2], lego: 64)")
    ap.add_argument("--white-bkgd", action="store_true", help="white background compositing (NeRF-synthetic scenes, BASELINE configs[2])")
    ap.add_argument("--no-normals", action="store_true",
                    help="calc_normal=False (SURVEY 8d config 2 asks for both): no nablas at the N sample points, no normals_volume")
    args = ap.parse_args()
    if args.workload == "stress5":
        return stress5(args)

    import torch
    import torch.distributed as dist
    from neumesh_amd import _lib
    from neumesh_amd.renderer import make_render_cfg, render_rays_fused
    from neumesh_amd.sharded import pack_outputs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    lib = _lib.load()

    mesh, model = build_scene(args.V, dev)
    model.mlp_precision = args.mlp_precision
    if args.samples < 8 or args.samples % 8:
        raise SystemExit("--samples must be a multiple of 8 (two halves, four up-sampling iterations)")
    cfg = make_render_cfg(calc_normal=not args.no_normals, N_samples=args.samples // 2, N_importance=args.samples // 2,
                          white_bkgd=args.white_bkgd)
    n_rays = args.H * args.W
    total_steps = args.warmup + args.steps
    from neumesh_amd import synthetic
    from neumesh_amd.rays import make_rays
    rays = []
    for s in range(total_steps):  # every rank builds the rays of ITS frame of the orbit on ITS GPU (nm_make_rays): resident before timing
        rays.append(make_rays(synthetic.orbit_pose(s * world + rank), synthetic.pinhole_intrinsics(args.H, args.W), args.H, args.W, dev))
    tables = model.field_tables()
    model.field_handle()
    gathered = torch.empty((world * n_rays, 8), dtype=torch.float32, device=dev) if world > 1 else None

    def step(i):
        ret = render_rays_fused(model, rays[i][0], rays[i][1], cfg, args.rayschunk or n_rays, tables=tables)
        if world > 1:
            packed, _ = pack_outputs(ret)
            dist.all_gather_into_tensor(gathered, packed)   # the path's only collective: final pixels
        return ret

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    rgb0 = None
    for i in range(args.warmup):
        ret = step(i)
        if i == 0 and rank == 0:
            rgb0 = ret["rgb"].cpu().numpy()
    if args.warmup == 0 and rank == 0 and args.cpu_rays > 0 and world == 1:
        rgb0 = None
    fence()
    lib.nm_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        ret = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel time inside the timed region (HIP events on the launch stream)
    kinds = {0: ("knn_distance", None), 1: ("geo_mlp", FLOP_GEO), 2: ("geo_mlp_tangent", FLOP_GEO + FLOP_TANGENT), 3: ("color_mlp", FLOP_COL)}
    prof = {}
    for k, (name, flop) in kinds.items():
        ms, n, u = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(lib.nm_profile_read(k, C.byref(ms), C.byref(n), C.byref(u)), "nm_profile_read")
        prof[name] = {"ms": ms.value, "launches": n.value, "points": u.value, "flop_per_point": flop}
    lib.nm_profile_enable(0)

    if rank == 0:
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))
            except Exception:
                traffic = None
        rays_total = world * n_rays * args.steps
        value = rays_total / elapsed
        split = args.mlp_precision == "f16x2"
        dom = max(("geo_mlp", "geo_mlp_tangent", "color_mlp"), key=lambda k: prof[k]["ms"])
        p = prof[dom]
        achieved = p["points"] * p["flop_per_point"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        mlp_flop = sum(prof[k]["points"] * prof[k]["flop_per_point"] for k in ("geo_mlp", "geo_mlp_tangent", "color_mlp"))
        mlp_ms = sum(prof[k]["ms"] for k in ("geo_mlp", "geo_mlp_tangent", "color_mlp"))
        kd = prof["knn_distance"]
        out = {
            "metric": f"rays/sec at {args.H}x{args.W}x{args.samples} samples (DTU scan63 shape, synthetic scene S-DTU)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_frame": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x2-split (22-bit operands, fp32 accumulate; K-NN and per-ray stages fp32)" if split else "f32", "data": "synthetic",
            "config": {"workload": f"S-DTU V={args.V} {args.H}x{args.W} rays/frame/GPU, {args.samples // 2}+{args.samples // 2} samples{', white background' if args.white_bkgd else ''}, bounded_near_far (256 probes), "
                                   + (f"calc_normal, per ray {256 + 3 * args.samples - 1} K-NN points ({256 + 2 * args.samples - 1} searched at most, {args.samples} reused), {2 * args.samples - 1} geometry-MLP evaluations with nablas (the reference's {args.samples} forward-only ones at the same points are the value rows of these) + {args.samples - 1} colour-MLP; probes between the first and last hit and mid-points of weight 0 are not evaluated"
                                    if not args.no_normals else f"calc_normal=False, per ray {256 + 3 * args.samples - 1} K-NN points ({256 + 2 * args.samples - 1} searched at most, {args.samples} reused), {args.samples} forward-only + {args.samples - 1} nabla geometry-MLP evaluations + {args.samples - 1} colour-MLP; probes between the first and last hit and mid-points of weight 0 are not evaluated"),
                       "rayschunk": args.rayschunk or n_rays, "parallelism": f"rays sharded: {world} GPU(s) x 1 frame per step, 1 all-gather of pixels"},
            "roofline": {"bound": "mfma", "kernel": ({"geo_mlp": "nm_geo_mlp_h_kernel<false>", "geo_mlp_tangent": "nm_geo_mlp_h_kernel<true>",
                                                      "color_mlp": "nm_col_mlp_h_kernel"} if split else
                                                     {"geo_mlp": "nm_geo_mlp_kernel<false>", "geo_mlp_tangent": "nm_geo_mlp_kernel<true>",
                                                      "color_mlp": "nm_col_mlp_kernel"})[dom],
                         # split-half mode executes 3 f16 MFMA products per algorithmic fp32 product
                         "achieved": achieved * (3.0 if split else 1.0), "peak": PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved * (3.0 if split else 1.0) / (PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS),
                         "algorithmic_tflops": achieved, "algorithmic_vs_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "mfma_dtype": "f16 (x3 products per fp32 product, fp32 accumulate)" if split else "f32",
                         "traffic": (traffic or {}).get({"geo_mlp": "geo_mlp", "geo_mlp_tangent": "geo_mlp_tangent", "color_mlp": "color_mlp"}[dom], {}).get("hbm_bytes_per_launch") if traffic else None,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" if traffic else None,
                         "avg_launch_ms": p["ms"] / max(p["launches"], 1), "launches": p["launches"],
                         "all_mlp_kernels_tflops": mlp_flop / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0,
                         "share_of_step_time": {k: prof[k]["ms"] / (elapsed * 1e3) for k in prof},
                         "knn_kernel": {"queries_per_s": kd["points"] / (kd["ms"] * 1e-3) if kd["ms"] > 0 else 0.0,
                                        "algorithmic_GBs": kd["points"] * KNN_BYTES_PER_QUERY / (kd["ms"] * 1e-3) / 1e9 if kd["ms"] > 0 else 0.0,
                                        "hbm_frac": (kd["points"] * KNN_BYTES_PER_QUERY / (kd["ms"] * 1e-3) / 1e9) / PEAK_HBM_GBS if kd["ms"] > 0 else 0.0}},
        }
        if world == 1 and args.cpu_rays > 0:
            try:
                rays0 = (rays[0][0].cpu().numpy(), rays[0][1].cpu().numpy())
                base, parity = cpu_baseline(mesh, model, args.H, args.W, args.cpu_rays, rgb0, rays0, samples=args.samples,
                                            white_bkgd=args.white_bkgd, calc_normal=not args.no_normals)
                out["cpu_baseline"] = base
                if parity:
                    out["parity_vs_oracle"] = parity
                out["speedup_vs_cpu_baseline"] = value / base["value"]
            except Exception as e:  # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
