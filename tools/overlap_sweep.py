"""tools/overlap_sweep.py [--quick] [--out gpurun_out/overlap_sweep.json]

Frame time of the bench frame (800x800x128, surface scene, normals) as a function of how the call is cut into ray chunks in flight:
chunk size x lanes (streams) x nm_render_cfg.overlap / knn_keep / mlp_prio (pull-form K-NN kernels that make room for the other chunks'
MLP kernels, csrc/nm_kernels.h).  Every variant's pixels are compared bit for bit with the one-call frame.  One process, one gpurun call:
box-to-box variance (5 %) does not enter the comparison."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "overlap_sweep.json"))
    ap.add_argument("--variants", default="", help="comma list chunk:lanes:overlap:keep:prio (overrides the built-in sweep)")
    args = ap.parse_args()
    import torch
    import bench
    from neumesh_amd import synthetic
    from neumesh_amd.rays import make_rays
    from neumesh_amd.renderer import make_render_cfg, render_rays_fused, release_workspaces
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    H = W = 800
    mesh, model = bench.build_scene(140_000, dev, scene="surf")
    intr = synthetic.pinhole_intrinsics(H, W)
    total = 1 + args.steps
    rays = [make_rays(synthetic.orbit_pose(s), intr, H, W, dev) for s in range(total)]
    cfg = make_render_cfg(calc_normal=True, N_samples=64, N_importance=64)
    tables = model.field_tables()
    model.field_handle()

    def frame(i, chunk):
        return render_rays_fused(model, rays[i][0], rays[i][1], cfg, chunk, tables=tables)

    def variant(chunk, lanes, overlap, keep, prio):
        os.environ["NEUMESH_RAYSCHUNK"] = "0"
        os.environ["NEUMESH_RENDER_STREAMS"] = str(lanes)
        os.environ["NEUMESH_OVERLAP"] = str(2 if (overlap and lanes == 1) else overlap)
        os.environ["NEUMESH_KNN_KEEP"] = str(keep)
        os.environ["NEUMESH_MLP_PRIO"] = str(prio)
        out0 = frame(0, chunk)
        torch.cuda.synchronize()
        lib.nm_profile_enable(1)
        t0 = time.perf_counter()
        for i in range(1, total):
            frame(i, chunk)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        prof = bench.read_prof(lib)
        lib.nm_profile_enable(0)
        variant.knn_ms = prof["knn_distance"]["ms"] / args.steps
        variant.mlp_ms = sum(prof[k]["ms"] for k in ("geo_mlp", "geo_mlp_tangent", "color_mlp")) / args.steps
        return ms, out0

    from neumesh_amd import _lib
    lib = _lib.load()
    variant(H * W, 1, 0, 0, 0)
    ms_ref, ref = variant(H * W, 1, 0, 0, 0)
    cfg.flags = _lib.RENDER_FORK_MID
    ms_fork, out = variant(H * W, 1, 0, 0, 0)
    cfg.flags = 0
    same = all(torch.equal(out[k], ref[k]) for k in ref)
    rows = [dict(chunk=H * W, lanes=1, overlap=0, keep=0, prio=0, ms=round(ms_ref, 2), identical=True),
            dict(chunk=H * W, lanes=1, overlap=0, keep=0, prio=0, ms=round(ms_fork, 2), identical=bool(same), note="NM_RENDER_FORK_MID")]
    print(f"one call, one stream: {ms_ref:.1f} ms; with the mid-point search on a side stream (NM_RENDER_FORK_MID): {ms_fork:.1f} ms {'identical' if same else 'PIXELS DIFFER'}", flush=True)
    if args.variants:
        todo = [tuple(int(x) for x in v.split(":")) for v in args.variants.split(",")]
    else:
        chunks = [65536, 131072] if args.quick else [65536, 106667, 160000, 213334]
        lanes_l = [2, 4] if args.quick else [2, 3, 4, 6]
        modes = [(0, 0, 0), (1, 1, 0), (1, 1, 2)] if args.quick else [(0, 0, 0), (1, 1, 0), (1, 1, 2), (1, 2, 0), (1, 2, 2), (1, 8, 0)]
        todo = [(c, l, *m) for c in chunks for l in lanes_l for m in modes if l <= -(-H * W // c)]
    for chunk, lanes, overlap, keep, prio in todo:
        try:
            ms, out = variant(chunk, lanes, overlap, keep, prio)
            same = all(torch.equal(out[k], ref[k]) for k in ref)
        except Exception as e:   # (a variant that does not fit the memory must not end the sweep)
            print(f"chunk {chunk} lanes {lanes} overlap {overlap}/{keep}/{prio}: {e}", flush=True)
            release_workspaces()
            continue
        rows.append(dict(chunk=chunk, lanes=lanes, overlap=overlap, keep=keep, prio=prio, ms=round(ms, 2), identical=bool(same)))
        rows[-1].update(knn_ms=round(variant.knn_ms, 2), mlp_ms=round(variant.mlp_ms, 2))
        print(f"chunk {chunk:7d} lanes {lanes} overlap {overlap} keep {keep} prio {prio}: {ms:7.1f} ms  (K-NN kernels {variant.knn_ms:.1f}, MLP kernels {variant.mlp_ms:.1f})  {'identical' if same else 'PIXELS DIFFER'}", flush=True)
        if chunk * lanes > 400_000:
            release_workspaces()
    ms_end, _ = variant(H * W, 1, 0, 0, 0)
    rows.append(dict(chunk=H * W, lanes=1, overlap=0, keep=0, prio=0, ms=round(ms_end, 2), identical=True, note="again, at the end"))
    best = min(rows, key=lambda r: r["ms"])
    print("best:", best, f"(one call: {ms_ref:.1f} / {ms_end:.1f} ms)", flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(dict(hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"), rows=rows, best=best), open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
