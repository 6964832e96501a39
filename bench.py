#!/usr/bin/env python
"""bench.py -- headline benchmark of the NeuMesh volumetric-render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torch.distributed environment re-executes itself under `torch.distributed.run`
(one rank per GPU, 127.0.0.1 rendezvous); launched by the driver under torch.distributed.run it
uses the environment it is given.

Workload (BASELINE.json configs[1], shape only -- there is no DTU data / checkpoint in the
environment, SURVEY.md section 8d scene S-DTU): V = 140 000-vertex prior mesh, 32-d geometry /
colour codes, W=256 MLPs.  --scene surf (default): the weights of synthetic.surface_mlp_state -- a field
WITH a surface (sdf = ds + a code-driven bump, s = 400; 23 % of the rays miss, 8 % graze, 59 % are opaque),
the scene whose reference render is pinned by tests/golden/render_v140k_surf.npz; --scene noise: the
default-initialised weights of rounds 1-2 (every ray opaque, s = 200; render_v140k_dtu.npz).  One STEP =
one 800x800 frame = 640 000 rays x (64 coarse + 64 importance) samples with bounded near/far (256
probes/ray) and normals, i.e. the kwargs get_model() hands render.py for
configs/neumesh_dtu_scan63.yaml.  Rays are resident in HBM before the timed region.  With N GPUs
every rank renders its own frame of the orbit per step (weak scaling: per-GPU work is fixed) and the
final pixels are all-gathered over RCCL -- the only collective of the path; --shard frame instead splits
ONE frame per step over the ranks by interleaved 32x32 tiles (BASELINE configs[3]; strong scaling).

Prints ONE JSON line (rank 0):
  value        rays/s of the whole job
  roofline     the dominant kernel, measured live with HIP events on the launch stream inside the
               timed region: achieved = ALGORITHMIC fp32 FLOP / time against the peak of the matrix
               pipe it runs on (frac); the issued-MFMA utilisation is a separate key
  cpu_baseline the CPU oracle (numpy + kd-tree K-NN) on a bounded ray sample of the same frame
  parity_vs_reference   the same frame's 1536 fixture rays against the imported reference's output
  config       besides the workload: scalar results of short runs (2 steps each, N = 1 only) of the variants the
               headline does not show -- data_independent_* (every probe and mid-point evaluated: the reference's
               work), fp32_*, f16_single_* (one f16 MFMA per product, with its error against the reference fixture),
               noise_scene_*, config5_* -- kept as scalars because the driver's record keeps scalars
  extra        the same runs in full + calc_normal=False, BASELINE config 3 / 4 shapes, the SURVEY 8f consumer rows
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import threading
import socket
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_GEO = 353_280          # geometry MLP forward, per point
FLOP_TANGENT = 271_360      # + forward-mode tangent (nabla)
FLOP_COL = 500_736          # colour MLP
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
DEFAULT_PRECISION = "f16x2s"    # the library's default MLP arithmetic (neumesh_amd/neumesh.py); rows of other modes are labelled
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: BF16/FP16 MFMA ~2.5 PF dense
PEAK_HBM_GBS = 8000.0
NOMINAL_CLOCK_MHZ = 2400.0      # the clock the dense MFMA peaks above are quoted at
KNN_BYTES_PER_QUERY = 76    # 12 in + 8*4 idx + 8*4 w  
PROFILE_TAG, PROFILE_TAG_PREVIOUS = "r06", "r05"          # profiles/<tag>_pmc_*.json: rocprofv3 --pmc passes of this command (tools/pmc_*.py)

MODEL_CFG = dict(D_density=3, D_color=4, W=256, geometry_dim=32, color_dim=32, multires_view=4, multires_d=8,
                 multires_fg=2, multires_ft=2, enable_nablas_input=True, speed_factor=10.0, learn_indicator_weight=False)
KINDS = {0: ("knn_distance", None), 1: ("geo_mlp", FLOP_GEO), 2: ("geo_mlp_tangent", FLOP_GEO + FLOP_TANGENT), 3: ("color_mlp", FLOP_COL)}


class _Mesh:
    def __init__(self, m):
        self.vertices, self.vertex_normals = m.vertices.astype(np.float64), m.vertex_normals.astype(np.float64)

    def compute_vertex_normals(self):
        return self


# This script sets the chunking of every run itself (whole frame in one call for the headline, half frames for the two-stream row, one
# 800x800 frame per chunk for config 4): the library's own policy -- the caller's rayschunk is only a lower bound, renderer._fused_chunk --
# is switched off so that each row measures what its label says.
os.environ.setdefault("NEUMESH_RAYSCHUNK", "0")
# The one-call frame's 40 GB workspace stays pooled between frames: handed back after every call (the library's policy above 24 GB, NEUMESH_WS_KEEP_GB) it can
# come back SPLIT by the small tensors allocated in between, the next frame then takes a fresh 40 GB from the driver, and on a box whose memory
# was just released by another process that costs 1-2 s inside ONE frame (round 5: first timed frame 1302-1918 ms, the rest 335;
# tools/stall_diag.py).  Nothing of a frame's work; the library's default chunks (20.6 GB per lane) are below the limit and never handed back.
os.environ.setdefault("NEUMESH_WS_KEEP_GB", "64")


def build_scene(V, device, seed=0, s_value=None, scene="surf"):
    """Scene S-DTU.  MLP weights: the set shared by every golden fixture
    (tests/golden/model_seed0.npz = the reference constructor under torch.manual_seed(0)), so that the
    benchmark scene is exactly the scene of tests/golden/render_v140k_dtu.npz; torch default init under
    `seed` if that file is missing."""
    import torch
    from neumesh_amd import MeshGrid, NeuMesh, synthetic
    mesh = synthetic.fibonacci_blob(V)
    torch.manual_seed(seed)
    model = NeuMesh(MeshGrid(_Mesh(mesh), device), **MODEL_CFG)
    wpath = os.path.join(ROOT, "tests", "golden", "model_seed0.npz")
    if scene == "trained":   # a TRAINED field (tools/train_field.py, 20 000 iterations of the reference's recipe on an analytic scene, s = 1000): the
        # checkpoint in utils/checkpoints.py layout that tests/golden/render_v140k_trained.npz is the imported reference's render of
        ck = os.path.join(ROOT, "tests", "golden", "trained_v140k.pt")
        if V != 140_000 or not os.path.exists(ck):
            raise SystemExit("scene 'trained' is tests/golden/trained_v140k.pt at V = 140000")
        model.load_state_dict(torch.load(ck, map_location="cpu")["model"])     # render.py:287-288
        return mesh, model.to(device).eval()
    if s_value is None:
        s_value = 400.0 if scene == "surf" else 200.0
    if os.path.exists(wpath):
        sd = dict(np.load(wpath).items())
        if scene == "surf":   # scene with a surface (sdf = ds + bump): tests/golden/render_v140k_surf.npz is the reference's render of it
            sd = synthetic.surface_mlp_state(sd, s_value=s_value)
        sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
        res = model.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
    elif scene == "surf":
        raise SystemExit("scene 'surf' is derived from tests/golden/model_seed0.npz, which is missing")
    with torch.no_grad():
        model.geometry_features.copy_(torch.from_numpy(synthetic.random_codes(V, 32, 1)))
        model.color_features.copy_(torch.from_numpy(synthetic.random_codes(V, 32, 2)))
        model.indicator_vector.copy_(torch.from_numpy(synthetic.noisy_indicator(mesh.vertex_normals, 3)))
        model.ln_s.fill_(float(np.log(s_value) / MODEL_CFG["speed_factor"]))
    return mesh, model.to(device).eval()


def frame_rays(frame, H, W):
    from neumesh_amd import synthetic
    return synthetic.camera_rays(synthetic.orbit_pose(frame), synthetic.pinhole_intrinsics(H, W), H, W)


def reference_baseline(mesh, model, H, W, n_rays, rays0, samples=128, white_bkgd=False, calc_normal=True):
    """The REFERENCE's own CPU PyTorch path (kind "reference") on a strided sample of frame 0's rays: the imported
    reference (oracle/refimport: /root/reference behind stub modules, FRNN replaced by the kd-tree + declared-arithmetic
    K-NN) with this scene's weights, torch threads = host cores.  Only where the reference tree exists (the build
    container; never on the GPU box).  Returns (baseline dict, rgb of the sample, ray indices)."""
    import torch
    from scipy.spatial import cKDTree
    from oracle import knn as oknn
    from oracle.refimport import harness
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    mlp = {k: v for k, v in state.items() if k not in ("geometry_features", "color_features", "indicator_vector")}
    ref_model, kw, renderer, _ = harness.build_reference(mesh, seed=0, mlp_state=mlp, s_value=float(model.forward_s()))
    import frnn as frnn_stub
    tree = cKDTree(mesh.vertices.astype(np.float64))
    frnn_stub.KNN_FN[0] = lambda q, v, K: oknn.knn_kdtree(q, v, K, tree=tree)
    o, d = rays0
    sel = np.linspace(0, H * W - 1, n_rays).astype(np.int64)
    kw = dict(kw)
    kw.update(rayschunk=n_rays, calc_normal=calc_normal, N_samples=samples // 2, N_importance=samples // 2, perturb=False, white_bkgd=white_bkgd)
    with torch.no_grad():
        renderer(torch.from_numpy(o[sel[:8]])[None], torch.from_numpy(d[sel[:8]])[None], detailed_output=False, **dict(kw, rayschunk=8))
        t = time.perf_counter()
        rgb, _, _ = renderer(torch.from_numpy(o[sel])[None], torch.from_numpy(d[sel])[None], detailed_output=False, **kw)
        dt = time.perf_counter() - t
    res = {"value": n_rays / dt, "unit": "rays/s", "cores": os.cpu_count(), "kind": "reference",
           "sample": f"{n_rays} rays strided over frame 0 of the same {H}x{W}x{samples} workload, {dt:.1f} s; the imported reference "
                     f"(models/renderer.py + neumesh.py on CPU torch, {torch.get_num_threads()} threads; FRNN stand-in: scipy cKDTree + declared fp32 re-rank)"}
    return res, rgb[0].numpy(), sel


def cpu_baseline(mesh, model, H, W, n_rays, rays0, samples=128, white_bkgd=False, calc_normal=True):
    """Oracle (CPU restatement of the reference, kind "port") on a strided sample of frame 0's rays.
    Returns (baseline dict, oracle rgb of the sample, ray indices)."""
    from oracle import field as ofield, knn as oknn, render as orender
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    orc = ofield.OracleField(mesh.vertices, state, ofield.FieldConfig(speed_factor=MODEL_CFG["speed_factor"]))
    from scipy.spatial import cKDTree
    tree = cKDTree(mesh.vertices.astype(np.float64))
    orc.knn_fn = lambda q, v, K: oknn.knn_kdtree(q, v, K, tree=tree)
    o, d = rays0
    sel = np.linspace(0, H * W - 1, n_rays).astype(np.int64)
    cfg = orender.RenderConfig(calc_normal=calc_normal, white_bkgd=white_bkgd, N_samples=samples // 2, N_importance=samples // 2)
    orender.render_rays(orc, o[sel[:8]], d[sel[:8]], cfg)  # warm caches / thread pools
    t = time.perf_counter()
    out = orender.render_rays(orc, o[sel], d[sel], cfg)
    dt = time.perf_counter() - t
    # the imported REFERENCE itself, timed where its tree exists (the build container): the committed record oracle/gen_golden.py surfsens wrote
    # (tests/golden/REPORT.json "reference_timing": the un-spied production call, median of 3) -- a record, not a literal
    rt = {}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "REPORT.json")) as f:
            rt = json.load(f).get("reference_timing", {})
    except (OSError, ValueError):
        pass
    res = {"value": n_rays / dt, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port",
           "reference_rays_per_s_build_container": rt.get("rays_per_s"), "reference_cores_build_container": rt.get("cores"),
           "reference_timing_record": rt or None,
           "sample": f"{n_rays} rays strided over frame 0 of the same {H}x{W}x{samples} workload, {dt:.1f} s; numpy fp32 oracle + "
                     f"scipy cKDTree candidates re-ranked with the declared fp32 arithmetic (BLAS/OpenMP threads = all cores; the same oracle as one "
                     f"single-threaded process per core reached 298 rays/s on 256 x 256 rays and 47 rays/s on 256 x 22 rays on this box type: it does not scale, so the one-process figure stands); "
                     f"the imported REFERENCE itself (kind 'reference', used automatically where /root/reference exists): reference_timing_record, "
                     f"from tests/golden/REPORT.json (oracle/gen_golden.py surfsens, build container)"}
    return res, out["rgb"], sel


def _psnr(a, b):
    mse = float(np.mean((np.asarray(a, np.float64) - np.asarray(b, np.float64)) ** 2))
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)


def consumer_rows(mesh, model, dev, H, W):
    """Timings of the rows SURVEY 8f marks "next" on the bench scene (reported under `extra`; each is pinned for parity by
    tests/test_gpu_parity.py): one training step through Trainer.forward + backward + Adam (models/trainer.py:50-117,
    train.py:176), the surface renderer (models/ray_casting.py:228-320), a frame through the texture-editing wrapper
    (editing/texture_neumesh/texture_neumesh.py:53-122)."""
    import torch
    from neumesh_amd import ray_casting as rc, synthetic
    from neumesh_amd.editing import TextureEditableNeuMesh
    from neumesh_amd.renderer import volume_render
    from neumesh_amd.trainer import Trainer
    out = {}

    def timed(fn, steps, warmup=1):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / steps

    pose, K = synthetic.orbit_pose(0), synthetic.pinhole_intrinsics(H, W)
    try:   # ---- training step: 512 random pixels of one view (the reference config's data.N_rays), eikonal + mask + regulariser on
        lw = {"img": 1.0, "eikonal": 0.1, "mask": 0.1, "indicator_reg": 0.1, "distill_density": 0.0, "distill_color": 0.0}
        trainer = Trainer(model, loss_weights=lw, teacher_model=None, device_ids=[dev.index or 0])
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=0.0)   # (the full update runs; a zero step keeps the timed steps on one and the same field)
        # (the view's images resident on the device, like every other timed input here; the 4x4 pose / intrinsics stay host tensors, as
        #  train.py's data loader hands them over: they go into the kernels by value)
        model_input = {"intrinsics": torch.from_numpy(np.asarray(K, np.float32))[None], "c2w": torch.from_numpy(np.asarray(pose, np.float32))[None],
                       "object_mask": torch.ones(1, H * W, dtype=torch.bool, device=dev)}
        gt = {"rgb": torch.full((1, H * W, 3), 0.5, device=dev)}
        kw = dict(N_nograd_samples=2048, N_upsample_iters=4, obj_bounding_radius=1.0, batched=True, perturb=True, white_bkgd=False,
                  bounded_near_far=True, calc_normal=True, H=H, W=W, N_samples=64, N_importance=64, rayschunk=4096)
        was_training = model.training
        saved = {k: v.detach().clone() for k, v in model.state_dict().items()}   # the steps below move the weights: put them back afterwards
        model.train()

        def step():
            opt.zero_grad(set_to_none=True)
            ret = trainer.forward({"data": {"N_rays": 512}}, None, model_input, gt, kw, 0, device=dev)
            ret["losses"]["total"].backward()
            opt.step()
        dt = timed(step, 8, 8)   # (the first steps pay for allocator growth and GEMM heuristics)
        backend = model.autograd_backend
        model.autograd_backend = "torch"      # the same step with the field evaluated by torch ops under autograd (what rounds 1-2 measured)
        dt_torch = timed(step, 4, 4)
        model.autograd_backend = backend
        model.load_state_dict(saved)
        model.train(was_training)
        for p_ in model.parameters():
            p_.grad = None
        out["train_step (512 rays x 128 samples of one view, img + eikonal + mask + indicator losses, forward + backward + Adam)"] = {
            "ms_per_step": dt * 1e3, "value": 512 / dt, "unit": "rays/s", "steps": 8,
            "field_backend": f"{backend}: nm_train_forward / nm_train_backward (closed-form reverse pass, GEMMs on the bf16 pipe with fp32 operands cut into three pieces)" if backend == "hip" else backend,
            "ms_per_step_torch_autograd_field": dt_torch * 1e3}
    except Exception as ex:
        out["train_step"] = {"error": str(ex)[-300:]}
    o, d = frame_rays(0, H, W)
    ro, rd = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    try:   # ---- surface renderer: first hit by 256 proposals + 8 secant steps, colour / normal at the hit
        with torch.no_grad():
            # the level set through the mesh (median field value at the vertices): the synthetic field's zero set need not cross the view
            tau = float(model.forward_density_only(torch.from_numpy(mesh.vertices[::7].astype(np.float32)).to(dev)).median())
            cfgs = dict(near=0.5, far=3.5, logit_tau=tau, fill_inf=False)
            hits = rc.surface_render(ro[None], rd[None], model, calc_normal=True, batched=True, rayschunk=1 << 17, ray_casting_algo="root_finding", ray_casting_cfgs=cfgs)[2]["mask_surface"]
            dt = timed(lambda: rc.surface_render(ro[None], rd[None], model, calc_normal=True, batched=True, rayschunk=1 << 17, ray_casting_algo="root_finding",
                                                 ray_casting_cfgs=cfgs), 2)
            dt_full = timed(lambda: rc.surface_render(ro[None], rd[None], model, calc_normal=True, batched=True, rayschunk=1 << 17, ray_casting_algo="root_finding",
                                                      ray_casting_cfgs=dict(cfgs, early_exit=False)), 1)
        out["surface_render (root finding: up to 256 proposals + 8 secant steps per ray, colour + normal at the hit)"] = {
            "ms_per_frame": dt * 1e3, "value": H * W / dt, "unit": "rays/s", "steps": 2, "rays_that_hit": float(hits.float().mean()),
            "ms_per_frame_all_256_proposals_evaluated": dt_full * 1e3}
    except Exception as ex:
        out["surface_render"] = {"error": str(ex)[-300:]}
    try:   # ---- texture editing: a tenth of the vertices painted from a second colour table, rendered by the staged renderer
        V = mesh.vertices.shape[0]
        g = torch.Generator().manual_seed(5)
        masks = (torch.rand(1, V, generator=g) < 0.1).to(dev)
        feats = (0.1 * torch.randn(V, model.color_features.shape[1], generator=g)).to(dev)
        edit = TextureEditableNeuMesh(model, [model], masks, feats)
        from neumesh_amd.renderer import make_render_cfg, render_rays_staged
        with torch.no_grad():
            dt = timed(lambda: volume_render(ro, rd, edit, calc_normal=False, perturb=False, detailed_output=False, rayschunk=H * W), 2)
            dt_staged = timed(lambda: render_rays_staged(edit, ro, rd, make_render_cfg(calc_normal=False), 1 << 17, 1 << 20), 1)
        out["texture_editing_render (TextureEditableNeuMesh, 10 % of the vertices painted: blend inside nm_render_rays)"] = {
            "ms_per_frame": dt * 1e3, "value": H * W / dt, "unit": "rays/s", "steps": 2,
            "staged_renderer_ms_per_frame (the wrapper's forward() per stage, as the reference drives it)": dt_staged * 1e3}
    except Exception as ex:
        out["texture_editing_render"] = {"error": str(ex)[-300:]}
    return out


def parity_blocks(gpu_rgb_frame0, H, W, V, oracle_rgb, sel, scene="surf", model=None):
    """(a) vs the committed REFERENCE output of the very same rays (fixture, 1536 rays of frame 0);
    (b) vs the oracle sample rendered for the CPU baseline."""
    out = {}
    fname = {"surf": "render_v140k_surf.npz", "trained": "render_v140k_trained.npz"}.get(scene, "render_v140k_dtu.npz")
    fpath = os.path.join(ROOT, "tests", "golden", fname)
    if gpu_rgb_frame0 is not None and os.path.exists(fpath):
        f = np.load(fpath)
        if int(f["V"]) == V and int(f["H"]) == H and int(f["W"]) == W:
            g = gpu_rgb_frame0[f["sel"]]
            err = np.abs(g - f["rgb"]).max(-1)
            se = f["self_err_1ulp"]
            out["parity_vs_reference"] = {
                "source": f"tests/golden/{fname}: the imported reference (CPU torch + declared-arithmetic K-NN) on these rays",
                "rays": int(len(err)), "psnr_db": _psnr(g, f["rgb"]), "max_abs_rgb": float(err.max()),
                "median_abs_rgb": float(np.median(err)), "frac_rays_within_1e-4": float((err <= 1e-4).mean()),
                "reference_self_sensitivity_1ulp": {"max_abs_rgb": float(se.max()), "frac_rays_within_1e-4": float((se <= 1e-4).mean())}}
            if model is not None and "d_all" in f.files:
                # everything behind the sampler on the REFERENCE's own sample depths (no last-bit sensitivity of the sample placement
                # involved): field + nablas + radiance + compositing of these 1536 rays against the reference's pixels
                try:
                    import torch
                    from neumesh_amd.renderer import make_render_cfg, render_at_depths
                    dev = next(model.parameters()).device
                    with torch.no_grad():
                        tail = render_at_depths(model, torch.from_numpy(f["rays_o"]).to(dev), torch.from_numpy(f["rays_d"]).to(dev),
                                                torch.from_numpy(f["d_all"]).to(dev), make_render_cfg(calc_normal=True))
                    et = np.abs(tail["rgb"].cpu().numpy() - f["rgb"]).max(-1)
                    out["parity_vs_reference"]["on_reference_depths"] = {
                        "max_abs_rgb": float(et.max()), "frac_rays_within_1e-4": float((et <= 1e-4).mean()),
                        "max_abs_depth": float(np.abs(tail["depth_volume"].cpu().numpy() - f["depth_volume"]).max()),
                        "max_abs_normals": float(np.abs(tail["normals_volume"].cpu().numpy() - f["normals_volume"]).max())}
                except Exception as ex:
                    out["parity_vs_reference"]["on_reference_depths"] = {"error": str(ex)}
    if gpu_rgb_frame0 is not None and oracle_rgb is not None:
        g = gpu_rgb_frame0[sel]
        err = np.abs(g - oracle_rgb).max(-1)
        out["parity_vs_oracle"] = {"rays": int(len(err)), "psnr_db": _psnr(g, oracle_rgb), "max_abs_rgb": float(err.max()),
                                   "median_abs_rgb": float(np.median(err)), "frac_rays_within_1e-4": float((err <= 1e-4).mean())}
    return out


def _load_profile(name):
    for tag in (PROFILE_TAG, PROFILE_TAG_PREVIOUS):   # (the round's own passes once they are committed; until then the previous round's)
        path = os.path.join(ROOT, "profiles", f"{tag}_{name}.json")
        try:
            return json.load(open(path)), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def read_prof(lib):
    from neumesh_amd import _lib
    prof = {}
    for k, (name, flop) in KINDS.items():
        ms, n, u = C.c_double(), C.c_int64(), C.c_int64()
        _lib.check(lib.nm_profile_read(k, C.byref(ms), C.byref(n), C.byref(u)), "nm_profile_read")
        prof[name] = {"ms": ms.value, "launches": n.value, "points": u.value, "flop_per_point": flop}
    return prof


def clock_under_load(lib, dev, work, micros=300, pause_s=0.02):
    """Shader clock (MHz) while `work()` -- which enqueues GPU work and returns -- is executing: a sampler thread calls nm_profile_clock on a
    stream of its own (one wave counting its clock against the 100 MHz counter for `micros` us) until the work has drained.
    Returns dict(mean, min, max, samples) or None."""
    import torch
    samples, stop = [], threading.Event()
    side = torch.cuda.Stream(device=dev)

    def sampler():
        torch.cuda.set_device(dev)
        mhz = C.c_float()
        while not stop.is_set():
            if lib.nm_profile_clock(micros, C.byref(mhz), C.c_void_p(side.cuda_stream)) == 0 and mhz.value > 0:
                samples.append(float(mhz.value))
            time.sleep(pause_s)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()                   # (the render call below blocks until its frame is done -- it polls the fp16-range flag -- so the sampler runs beside it)
    time.sleep(0.05)
    samples.clear()              # (what was measured before the work started does not count)
    work()
    torch.cuda.current_stream(dev).synchronize()
    stop.set()
    th.join(timeout=5)
    if len(samples) > 2:
        samples = samples[:-1]   # (the last one may have run after the work ended)
    if not samples:
        return None
    return {"mean": sum(samples) / len(samples), "min": min(samples), "max": max(samples), "samples": len(samples)}


def stress5_run(args, dev, world, rank, steps, warmup):
    """BASELINE config 5, the HBM-bound case of the path: V = 1 000 000 vertices, one 256-d
    vertex feature table, kernels = K-NN + gather-interpolate only (nm_distance_interpolate), queries =
    the 4096x4096 rays of a frame, one point per ray where it meets the surface shell.  One step = one
    such frame, in slabs of 256 image rows (the 1 KiB/query output of a slab is 1 GiB)."""
    import torch
    import torch.distributed as dist
    from neumesh_amd import _lib, synthetic
    from neumesh_amd.mesh_grid import MeshGrid
    from neumesh_amd.rays import make_rays
    lib = _lib.load()
    V, dim, H, W, slab = 1_000_000, 256, 4096, 4096, 256
    mesh = synthetic.fibonacci_blob(V)
    grid = MeshGrid(_Mesh(mesh), dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    table = torch.randn((V, dim), generator=gen, device=dev)
    ind = grid.vertex_normals.contiguous()
    K = synthetic.pinhole_intrinsics(H, W)
    total = warmup + steps
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

    for i in range(warmup):
        step(i)
    fence()
    lib.nm_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(warmup, total):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms, n, u = C.c_double(), C.c_int64(), C.c_int64()
    _lib.check(lib.nm_profile_read(0, C.byref(ms), C.byref(n), C.byref(u)), "nm_profile_read")
    lib.nm_profile_enable(0)
    bytes_q = 12 + 8 * dim * 4          # query + 8 gathered rows (the 4*dim-byte output row is extra)
    per_launch_q = u.value / max(n.value, 1)
    avg_ms = ms.value / max(n.value, 1)
    achieved = per_launch_q * bytes_q / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, tsrc = _load_profile("pmc_traffic_stress5")
    hbm_per_launch = (traffic or {}).get("knn_distance", {}).get("hbm_bytes_per_launch")
    return {
        "metric": "K-NN + gather-interpolate queries/sec, 1M-vertex mesh x 256-d features, 4096x4096 rays (BASELINE config 5)",
        "value": world * H * W * steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"stress5: V={V}, {dim}-d table ({V * dim * 4 / 2**30:.2f} GiB), {H}x{W} queries per step per GPU in slabs of {slab} rows",
                   "parallelism": f"queries sharded: {world} GPU(s) x 1 frame per step, no collective"},
        "roofline": {"bound": "hbm", "kernel": "nm_distance_kernel<false> (K-NN + weights + 8-row gather-interpolate)",
                     "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": achieved / PEAK_HBM_GBS,
                     "bytes_per_query": bytes_q, "written_bytes_per_query_not_counted": dim * 4 + 4,
                     "avg_launch_ms": avg_ms, "launches": n.value,
                     "traffic": hbm_per_launch, "traffic_source": tsrc,
                     "measured_hbm_GBs": (hbm_per_launch / (avg_ms * 1e-3) / 1e9) if (hbm_per_launch and avg_ms > 0) else None,
                     "note": "algorithmic bytes assume every gathered row comes from HBM; neighbouring queries share rows, so the "
                             "measured HBM traffic (`traffic`) is far below it and the kernel is not HBM-bound in practice"}}


# The driver keeps the first 24 keys of `config`: scalars first, in the order a reader needs them; the
# descriptive strings (scene, work strategy, reference work per ray, parallelism ...) behind.
CONFIG_KEY_ORDER = [
    "workload",
    "parity_on_reference_depths_max_abs_rgb", "parity_max_abs_rgb_vs_reference", "parity_median_abs_rgb_vs_reference", "parity_frac_rays_within_1e-4",
    "reference_self_1ulp_frac_rays_within_1e-4",
    "data_independent_rays_per_s", "data_independent_ms_per_frame",
    "fp32_rays_per_s", "fp32_ms_per_frame", "fp32_frac_of_fp32_mfma_peak",
    "two_accumulator_f16x2_rays_per_s", "two_accumulator_f16x2_max_abs_rgb_vs_reference",
    "f16col_rays_per_s", "f16col_max_abs_rgb_vs_reference", "f16col_frac_rays_within_1e-4",
    "f16_single_rays_per_s", "f16_single_max_abs_rgb_vs_reference", "f16_single_frac_rays_within_1e-4",
    "config5_frac_algorithmic_of_hbm_peak", "config5_frac_measured_hbm_of_peak", "config5_queries_per_s",
    "train_step_ms_512_rays", "noise_scene_rays_per_s",
]


def order_config(cfg: dict) -> dict:
    head = {k: cfg[k] for k in CONFIG_KEY_ORDER if k in cfg}
    head.update({k: v for k, v in cfg.items() if k not in head})
    return head


def line_guard(out, extra, budget_s):
    """(emit, timer) for the ONE line of the contract.  emit(note=None) prints `out` (+ `extra` as out["extra"]) exactly once; timer is an
    unstarted threading.Timer that, when it fires after budget_s seconds, emits the line with the rows finished so far and ends the process
    with status 0 -- so a stalled row after the headline cannot cost the line (tests/test_host.py runs this against a sleeping main thread)."""
    printed = threading.Lock()

    def emit(note=None):
        if not printed.acquire(blocking=False):
            return False
        for _attempt in range(5):       # (the main thread may be adding a row at this very moment)
            try:
                if isinstance(out.get("config"), dict):
                    out["config"] = order_config(out["config"])
                snap = dict(extra)
                if note:
                    snap["_watchdog"] = note
                line = json.dumps(dict(out, extra=snap) if snap else out, default=str)
                break
            except RuntimeError:
                time.sleep(0.01)
        else:
            line = json.dumps({k: v for k, v in out.items() if k != "extra"}, default=str)
        print(line, flush=True)
        return True

    def watchdog():
        if emit(f"the rows after the headline did not finish within {budget_s} s: line printed with the rows completed so far"):
            os._exit(0)
    timer = threading.Timer(budget_s, watchdog)
    timer.daemon = True
    return emit, timer


def self_spawn(args):
    """`python bench.py --gpus N` outside torch.distributed.run: re-exec under it (one rank per GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execvpe(cmd[0], cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["frame", "stress5"], default="frame",
                    help="frame = BASELINE configs[1], the headline 800x800x128 render (default); stress5 = BASELINE configs[4], "
                         "the HBM-bound K-NN + 256-d gather stress (a second roofline, not the headline metric)")
    ap.add_argument("--scene", choices=["surf", "noise", "trained"], default="surf",
                    help="surf = MLP weights with a surface (synthetic.surface_mlp_state: sdf = ds + code-driven bump, s = 400; rays miss / graze / "
                         "hit); noise = the default-initialised weights of rounds 1-2 (every ray opaque, s = 200); trained = the checkpoint "
                         "tests/golden/trained_v140k.pt (the reference's training recipe on an analytic scene, s = 1000)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--H", type=int, default=800)
    ap.add_argument("--W", type=int, default=800)
    ap.add_argument("--V", type=int, default=140_000)
    ap.add_argument("--rayschunk", type=int, default=0,
                    help="rays per nm_render_rays call; 0 = the whole frame in one call (56 KB of workspace per ray: 36 GB for 800x800)")
    ap.add_argument("--mlp-precision", choices=["f16x2", "f16", "fp32", "f16x2+f16col", "f16x2s", "f16x2s+f16col"], default=DEFAULT_PRECISION,
                    help="MLP arithmetic: split-half f16 MFMA (default 'f16x2s': one accumulator; 'f16x2': two), single-product f16 MFMA "
                         "(reduced precision, error-quantified) or fp32 MFMA")
    ap.add_argument("--cpu-rays", type=int, default=1536, help="rays of the CPU-baseline sample (0 disables)")
    ap.add_argument("--samples", type=int, default=128, help="samples per ray, half coarse / half importance (BASELINE configs[2], lego: 64)")
    ap.add_argument("--white-bkgd", action="store_true", help="white background compositing (NeRF-synthetic scenes, BASELINE configs[2])")
    ap.add_argument("--no-normals", action="store_true",
                    help="calc_normal=False (SURVEY 8d config 2 asks for both): no nablas at the N sample points, no normals_volume")
    ap.add_argument("--data-independent", action="store_true",
                    help="evaluate every probe and every mid-point (NM_RENDER_FULL_PROBES | NM_RENDER_NO_ZERO_SKIP): the work the reference always does")
    ap.add_argument("--extras-budget", type=float, default=240.0, help="seconds the rows after the headline may take before the line is printed without the rest")
    ap.add_argument("--no-extras", action="store_true", help="skip the short variant runs reported under `config` / `extra`")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1: nccl = RCCL over xGMI (default); gloo = test mode for boxes with fewer GPUs than "
                         "ranks (ranks share devices round-robin, the collectives are staged through host memory)")
    ap.add_argument("--shard", choices=["frames", "frame"], default="frames",
                    help="multi-GPU partition: frames = every rank renders its own frame per step (weak scaling, default); frame = ONE "
                         "frame per step split over the ranks by interleaved 32x32 tiles (BASELINE configs[3]; strong scaling)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)

    import torch
    import torch.distributed as dist
    from neumesh_amd import _lib
    from neumesh_amd.renderer import make_render_cfg, render_rays_fused
    from neumesh_amd.sharded import _all_gather_rows, pack_outputs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.backend == "gloo":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    lib = _lib.load()

    if args.workload == "stress5":
        out = stress5_run(args, dev, world, rank, args.steps, args.warmup)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return

    if args.samples < 8 or args.samples % 8:
        raise SystemExit("--samples must be a multiple of 8 (two halves, four up-sampling iterations)")
    mesh, model = build_scene(args.V, dev, scene=args.scene)
    from neumesh_amd import synthetic
    from neumesh_amd.rays import make_rays
    from neumesh_amd.sharded import all_gather_rows_async, render_frame_sharded_async
    n_rays = args.H * args.W
    intr = synthetic.pinhole_intrinsics(args.H, args.W)
    one_frame = world > 1 and args.shard == "frame"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(steps, warmup, precision=None, samples=128, normals=True, white=False, flags=0, keep_frame0=False, gather=True, hw=None,
            weight_eps=0.0, chunk=None, mdl=None, library_policy=False):
        """warmup + `steps` timed frames of one variant; returns (elapsed s [max over ranks], profile dict, rgb of frame 0 or None,
        rays of frame 0 or None, per-rank seconds up to the end of the rank's own work)."""
        m = mdl or model
        m.mlp_precision = precision or args.mlp_precision
        cfg = make_render_cfg(calc_normal=normals, N_samples=samples // 2, N_importance=samples // 2, white_bkgd=white, flags=flags, weight_eps=weight_eps)
        total = warmup + steps
        H, W = hw or (args.H, args.W)
        r_intr = synthetic.pinhole_intrinsics(H, W) if hw else intr
        tables = m.field_tables()
        m.field_handle()
        rc = chunk or args.rayschunk or n_rays

        def render(ro, rd):
            if library_policy:   # what render.py gets: its rayschunk = 4096 (a lower bound here), the chunking chosen by renderer._fused_chunk
                saved = os.environ.pop("NEUMESH_RAYSCHUNK", None)
                try:
                    return render_rays_fused(m, ro, rd, cfg, 4096, tables=tables)
                finally:
                    if saved is not None:
                        os.environ["NEUMESH_RAYSCHUNK"] = saved
            return render_rays_fused(m, ro, rd, cfg, rc, tables=tables)   # hw frames: chunks of one headline frame

        if one_frame:   # ONE frame per step over all ranks: every rank builds and renders the rays of its interleaved tiles
            poses = [synthetic.orbit_pose(s) for s in range(total)]
            rays = None
        else:           # every rank builds the rays of ITS frame of the orbit on ITS GPU (nm_make_rays): resident before timing
            rays = [make_rays(synthetic.orbit_pose(s * world + rank), r_intr, H, W, dev) for s in range(total)]
        do_gather = world > 1 and gather and not one_frame

        # The path's only collective -- final pixels -- is POSTED after a frame's render and waited for after the NEXT frame's kernels have been
        # queued (round 6: the renderer returns without a host sync), so frame i's all-gather and assembly run beside frame i + 1's kernels.
        pending = [None]

        def drain():
            if pending[0] is not None:
                pending[0]()
                pending[0] = None

        def step(i):
            if one_frame:
                nxt = render_frame_sharded_async(render, poses[i], r_intr, H, W, dev)
                prev, pending[0] = pending[0], nxt
                return prev() if prev is not None else None
            ret = render(rays[i][0], rays[i][1])
            if do_gather:
                packed, _ = pack_outputs(ret)
                nxt = all_gather_rows_async(packed, world).result   # ([world * H * W, 5 or 8] on every rank)
                prev, pending[0] = pending[0], nxt
                if prev is not None:
                    prev()
            return ret

        rgb0 = None
        # The warm-up runs exactly what the timed region runs, the in-stream kernel timing included: on a cold box the first use of the event
        # path costs the HIP runtime about a second ONCE (measured round 5: first timed frame 1302 ms, the next two 335), which does not
        # belong to a frame.  nm_profile_enable(1) before the timed region clears what the warm-up logged.
        lib.nm_profile_enable(1)
        rgb0_dev = None
        if total > 0 and not one_frame and rays:
            step(0)              # set-up, not a warm-up step of the contract: the first frame of a process takes the workspace from the driver
            drain()
            fence()              # (and whatever else is first-use); the W warm-up steps and the K timed steps follow
        for i in range(warmup):
            ret = step(i)
            if i == 0 and keep_frame0 and one_frame:   # (pipelined form: frame 0 would come back one step later)
                ret, pending[0] = pending[0](), None
            if i == 0 and keep_frame0 and rank == 0 and ret is not None:
                rgb0_dev = ret["rgb"].clone()   # (brought to the host AFTER the timed region: on a cold box a device-to-host copy here was followed by
                                                #  a 1-2 s stall inside the next frame -- tools/stall_diag.py, round 5 -- which is no part of a frame)
        drain()
        fence()
        lib.nm_profile_enable(1)
        stamps = []
        t0 = time.perf_counter()
        for i in range(warmup, total):
            step(i)
            stamps.append(time.perf_counter() - t0)   # (host time at which step i's call returned: no synchronisation added)
        drain()                            # the last frame's collective + assembly belong to the timed region
        torch.cuda.synchronize()
        own = time.perf_counter() - t0     # this rank's own work (+ the collectives it took part in), before the closing barrier
        run.last_step_returns_ms = [round(x * 1e3, 1) for x in stamps]
        fence()
        elapsed = time.perf_counter() - t0
        per_rank = [own]
        if world > 1:
            tdev = dev if args.backend == "nccl" else torch.device("cpu")
            t = torch.tensor([elapsed], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            allr = torch.zeros(world, dtype=torch.float64, device=tdev)
            dist.all_gather_into_tensor(allr, torch.tensor([own], dtype=torch.float64, device=tdev))
            per_rank = [float(x) for x in allr.tolist()]
        prof = read_prof(lib)
        lib.nm_profile_enable(0)
        if rgb0_dev is not None:
            rgb0 = rgb0_dev.cpu().numpy()
        return elapsed, prof, rgb0, (rays[0] if (keep_frame0 and rays) else None), per_rank

    head_flags = (_lib.RENDER_FULL_PROBES | _lib.RENDER_NO_ZERO_SKIP) if args.data_independent else 0
    elapsed, prof, rgb0, rays0, per_rank = run(args.steps, args.warmup, precision=args.mlp_precision, samples=args.samples,
                                               normals=not args.no_normals, white=args.white_bkgd, flags=head_flags, keep_frame0=True)

    def mlp_summary(prof, precision):
        split = precision != "fp32"
        dom = max(("geo_mlp", "geo_mlp_tangent", "color_mlp"), key=lambda k: prof[k]["ms"])
        p = prof[dom]
        alg = p["points"] * p["flop_per_point"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        peak = PEAK_F16_MFMA_TFLOPS if split else PEAK_FP32_MFMA_TFLOPS
        return dom, p, alg, peak, split

    if rank == 0:
        frames_per_step = 1 if one_frame else world
        value = frames_per_step * n_rays * args.steps / elapsed
        dom, p, alg, peak, split = mlp_summary(prof, args.mlp_precision)
        products = 3.0 if args.mlp_precision.startswith("f16x2") else 1.0   # (of the geometry kernels, which dominate; '+f16col': the colour kernel issues 1)
        mlp_flop = sum(prof[k]["points"] * prof[k]["flop_per_point"] for k in ("geo_mlp", "geo_mlp_tangent", "color_mlp"))
        mlp_ms = sum(prof[k]["ms"] for k in ("geo_mlp", "geo_mlp_tangent", "color_mlp"))
        kd = prof["knn_distance"]
        traffic, tsrc = _load_profile("pmc_traffic")
        mfma_pmc, msrc = _load_profile("pmc_mfma")
        knn_pmc, ksrc = _load_profile("pmc_knn")
        issue = None
        if knn_pmc and "knn_probe_bounds" in knn_pmc and "knn_distance" in knn_pmc and kd["ms"] > 0:
            # Issue-bound roofline of the K-NN kernels: vector instructions (wave-wide) the kernels executed per
            # searched point, from the committed SQ counter pass of this same workload (one probe launch per frame there), against the
            # rate the 1024 SIMDs can issue them: a wave64 fp32 vector instruction occupies a 16-lane SIMD for 4 cycles.
            frames_p = max(int(knn_pmc["knn_probe_bounds"].get("launches", 1)), 1)
            valu = sum(knn_pmc[k]["counters"]["SQ_INSTS_VALU"] for k in ("knn_probe_bounds", "knn_distance")) / frames_p
            salu = sum(knn_pmc[k]["counters"]["SQ_INSTS_SALU"] for k in ("knn_probe_bounds", "knn_distance")) / frames_p
            pts_f = kd["points"] / max(args.steps, 1)
            ms_f = kd["ms"] / max(args.steps, 1)
            valu_peak = 256 * 4 * 2.4e9 / 4 / 1e9
            issue = {"bound": "vector-instruction issue", "valu_wave_instructions_per_searched_point": valu / pts_f,
                     "salu_wave_instructions_per_searched_point": salu / pts_f,
                     "achieved": valu / (ms_f * 1e-3) / 1e9, "peak": valu_peak, "unit": "G wave-instructions/s (VALU)", "frac": valu / (ms_f * 1e-3) / 1e9 / valu_peak,
                     "note": "instruction counts from the profile named in issue_pmc_source (same workload), time from this run"}
        nabla_k, fixed_k = {"geo_mlp": "false", "geo_mlp_tangent": "true"}, "true"
        kname = ({"geo_mlp": "nm_geo_mlp_h2_kernel<false,true,NP>", "geo_mlp_tangent": "nm_geo_mlp_h2_kernel<true,true,NP>", "color_mlp": "nm_col_mlp_h2_kernel<true,NP>"} if split else
                 {"geo_mlp": "nm_geo_mlp_kernel<false>", "geo_mlp_tangent": "nm_geo_mlp_kernel<true>", "color_mlp": "nm_col_mlp_kernel"})[dom]
        np_geo = "6" if args.mlp_precision.startswith("f16x2s") else "3" if args.mlp_precision.startswith("f16x2") else "1"
        kname = kname.replace("NP", "1" if (dom == "color_mlp" and args.mlp_precision.endswith("+f16col")) else np_geo)
        searched_per_s = kd["points"] / (kd["ms"] * 1e-3) if kd["ms"] > 0 else 0.0
        n_frames = args.steps    # frames THIS rank's profile saw (its share of each under --shard frame)
        rays_here = (n_rays / world) if one_frame else n_rays
        knn_ref_per_ray = 256 + 3 * args.samples - 1                    # K-NN points the reference searches per ray
        mid_per_ray = prof["color_mlp"]["points"] / max(n_frames * rays_here, 1)
        strategy = ("every probe and every mid-point evaluated (data-independent work, as the reference)" if args.data_independent else
                    "probes between first/last hit and zero-weight mid-points skipped (bit-identical)")
        dtype = {"f16x2": "f16x2-split (22-bit operands, fp32 accumulate; K-NN and per-ray stages fp32)",
                 "f16x2s": "f16x2-split, one accumulator (operands h1 + unscaled residual half: 2^-25 absolute, fp32 accumulate; K-NN and per-ray stages fp32)",
                 "f16x2+f16col": "f16x2-split geometry network + f16 single-product colour network (error-quantified, see config.parity_*)",
                 "f16x2s+f16col": "f16x2-split one-accumulator geometry network + f16 single-product colour network (error-quantified, see config.parity_*)",
                 "f16": "f16 single product (11-bit operands, fp32 accumulate): reduced precision, see config.f16_single_*", "fp32": "f32"}[args.mlp_precision]
        out = {
            "metric": f"rays/sec at {args.H}x{args.W}x{args.samples} samples (DTU scan63 shape, synthetic scene S-DTU, {'with a surface' if args.scene == 'surf' else 'default-init noise field'})",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_frame": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if one_frame else "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"S-DTU/{args.scene} V={args.V} {args.H}x{args.W}x{args.samples // 2}+{args.samples // 2}, bounded near/far, normals={not args.no_normals}"
                                   f"{', white bkgd' if args.white_bkgd else ''}",
                       "scene": ("surf: sdf = ds + code-driven bump, s = 400 (synthetic.surface_mlp_state); tests/golden/render_v140k_surf.npz" if args.scene == "surf"
                                 else "noise: default-init MLP weights, s = 200; tests/golden/render_v140k_dtu.npz"),
                       "work_strategy": strategy,
                       "reference_work_per_ray": f"{knn_ref_per_ray} K-NN points, {2 * args.samples - 1} geometry-MLP + {args.samples - 1} colour-MLP evaluations",
                       "knn_points_searched_per_ray": kd["points"] / max(n_frames * rays_here, 1),
                       "knn_points_searched_frac_of_reference": kd["points"] / max(n_frames * rays_here, 1) / knn_ref_per_ray,
                       "midpoints_evaluated_per_ray": mid_per_ray, "midpoints_zero_weight_frac": 1.0 - mid_per_ray / (args.samples - 1),
                       "rayschunk": args.rayschunk or n_rays,
                       "parallelism": (f"one frame per step over {world} GPU(s): interleaved 32x32 pixel tiles, 1 all-gather of pixels" if one_frame else
                                       f"rays sharded: {world} GPU(s) x 1 frame per step, 1 all-gather of pixels"),
                       "world_size": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else "none",
                       "per_rank_ms_per_step_min": min(per_rank) / args.steps * 1e3, "per_rank_ms_per_step_max": max(per_rank) / args.steps * 1e3},
            "roofline": {"bound": "mfma", "kernel": kname,
                         # achieved = ALGORITHMIC fp32 flops of the layer products / measured kernel time
                         "achieved": alg, "peak": peak, "unit": "TFLOP/s", "frac": alg / peak,
                         "mfma_dtype": ("f16 (3 MFMA products per fp32 product: split-half operands, fp32 accumulate)" if products == 3.0 else
                                        "f16 (1 MFMA product)" if split else "f32"),
                         # what the matrix pipe actually executes (3x the algorithmic flops in split-half mode)
                         "issued_tflops": alg * products, "issued_frac_of_pipe_peak": alg * products / peak,
                         "algorithmic_vs_fp32_mfma_peak": alg / PEAK_FP32_MFMA_TFLOPS,
                         "mfma_busy_pmc": (mfma_pmc or {}).get(dom), "mfma_busy_source": msrc,
                         "traffic": (traffic or {}).get(dom, {}).get("hbm_bytes_per_launch"), "traffic_source": tsrc,
                         "algorithmic_bytes_per_launch": (p["points"] / max(p["launches"], 1)) * (160 if dom != "color_mlp" else 156),
                         "avg_launch_ms": p["ms"] / max(p["launches"], 1), "launches": p["launches"],
                         "points_per_launch": p["points"] / max(p["launches"], 1),
                         "all_mlp_kernels_tflops": mlp_flop / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0,
                         "share_of_step_time": {k: prof[k]["ms"] / (elapsed * 1e3) for k in prof}},
            "per_rank_ms_per_step": [x / args.steps * 1e3 for x in per_rank],
            "step_returns_ms": getattr(run, "last_step_returns_ms", None),   # host clock when each timed step's call returned, from the start of the timed region
            "knn_kernel": {"kernels": "nm_distance_kernel<chain> + nm_probe_bounds_kernel",
                           "bound": "instruction issue + scalar-load latency (index is L2/scalar-cache resident; not HBM)",
                           "searched_points_per_s": searched_per_s, "searched_points_per_frame": kd["points"] / max(n_frames, 1),
                           "ms_per_frame": kd["ms"] / max(n_frames, 1),
                           "algorithmic_GBs_at_76B_per_query": searched_per_s * KNN_BYTES_PER_QUERY / 1e9,
                           "hbm_frac_at_76B_per_query": searched_per_s * KNN_BYTES_PER_QUERY / 1e9 / PEAK_HBM_GBS,
                           "issue_roofline": issue, "issue_pmc": knn_pmc, "issue_pmc_source": ksrc,
                           "traffic": (traffic or {}).get("knn_distance", {}).get("hbm_bytes_per_launch")},
        }
        cfgd = out["config"]
        extra = {}
        if world == 1:   # the clock the chip holds under this very load (the pipe peaks above assume the nominal 2.4 GHz)
            try:
                m_ = model
                cfg_c = make_render_cfg(calc_normal=not args.no_normals, N_samples=args.samples // 2, N_importance=args.samples // 2, white_bkgd=args.white_bkgd, flags=head_flags)
                ro_c, rd_c = make_rays(synthetic.orbit_pose(0), intr, args.H, args.W, dev)
                tb_c = m_.field_tables()
                clk = clock_under_load(lib, dev, lambda: [render_rays_fused(m_, ro_c, rd_c, cfg_c, args.rayschunk or n_rays, tables=tb_c) for _ in range(2)])
                if clk:
                    out["roofline"]["shader_clock_mhz_under_load"] = clk
                    out["roofline"]["frac_at_measured_clock"] = alg / (peak * clk["mean"] / NOMINAL_CLOCK_MHZ)
                    out["roofline"]["clock_note"] = (f"peak = dense MFMA rate at the nominal {NOMINAL_CLOCK_MHZ:.0f} MHz; under this workload the shader clock is "
                                                     f"{clk['mean']:.0f} MHz (nm_profile_clock sampled beside two frames), i.e. the pipe's own ceiling here is "
                                                     f"{peak * clk['mean'] / NOMINAL_CLOCK_MHZ:.0f} TFLOP/s")
            except Exception as ex:
                out["roofline"]["shader_clock_mhz_under_load"] = {"error": str(ex)}
        # Everything from here on is context beside the headline (variant rows, consumers of the path, the CPU baseline).  A watchdog makes sure
        # the ONE line of the contract is printed even if one of those rows should stall: after --extras-budget seconds it prints the line with the
        # rows finished so far and ends the process.  (Blocking HIP calls release the interpreter lock, so the thread runs.)
        emit, timer = line_guard(out, extra, args.extras_budget)
        if world == 1:
            timer.start()
        fixture = None
        fx_path = os.path.join(ROOT, "tests", "golden", "render_v140k_surf.npz" if args.scene == "surf" else "render_v140k_dtu.npz")
        if os.path.exists(fx_path):
            fixture = np.load(fx_path)
            if not (int(fixture["V"]) == args.V and int(fixture["H"]) == args.H and int(fixture["W"]) == args.W):
                fixture = None
        if world == 1 and not args.no_extras and not args.data_independent and args.samples == 128 and not args.no_normals and not args.white_bkgd and args.mlp_precision == DEFAULT_PRECISION:
            name_of_last_short = [None]

            def short(name, **kw):
                name_of_last_short[0] = name
                try:
                    e, pr, img, _, _ = run(2, 1, **kw)
                    run.last_rgb0 = img
                    d, pp, a, pk, sp = mlp_summary(pr, kw.get("precision", args.mlp_precision))
                    nr = kw["hw"][0] * kw["hw"][1] if "hw" in kw else n_rays
                    extra[name] = {"value": nr * 2 / e, "unit": "rays/s", "ms_per_frame": e / 2 * 1e3, "steps": 2,
                                   "dominant_kernel": d, "achieved_tflops_algorithmic": a, "frac_of_pipe_peak": a / pk,
                                   "knn_ms_per_frame": pr["knn_distance"]["ms"] / 2, "knn_searched_per_frame": pr["knn_distance"]["points"] / 2,
                                   "mlp_points_per_frame": {k: pr[k]["points"] / 2 for k in ("geo_mlp", "geo_mlp_tangent", "color_mlp")}}
                    if img is not None and rgb0 is not None and img.shape == rgb0.shape:   # same frame 0 as the headline run
                        extra[name]["max_abs_rgb_vs_headline_frame"] = float(np.abs(img - rgb0).max())
                    if img is not None and fixture is not None and "mdl" not in kw:
                        g = img[fixture["sel"]]
                        err = np.abs(g - fixture["rgb"]).max(-1)
                        extra[name]["vs_reference_fixture"] = {"max_abs_rgb": float(err.max()), "median_abs_rgb": float(np.median(err)),
                                                               "frac_rays_within_1e-4": float((err <= 1e-4).mean()), "psnr_db": _psnr(g, fixture["rgb"])}
                    return extra[name]
                except Exception as ex:  # a variant must never sink the headline
                    extra[name] = {"error": str(ex)}
                    return {}
            r = short("data_independent_frame (every probe + every mid-point evaluated: the reference's work)",
                      flags=_lib.RENDER_FULL_PROBES | _lib.RENDER_NO_ZERO_SKIP)
            cfgd["data_independent_rays_per_s"], cfgd["data_independent_ms_per_frame"] = r.get("value"), r.get("ms_per_frame")
            r = short("mlp_precision_fp32 (fp32-input MFMA)", precision="fp32", keep_frame0=True)
            cfgd["fp32_rays_per_s"], cfgd["fp32_ms_per_frame"] = r.get("value"), r.get("ms_per_frame")
            fxr = r.get("vs_reference_fixture", {})
            cfgd["fp32_max_abs_rgb_vs_reference"], cfgd["fp32_frac_rays_within_1e-4"] = fxr.get("max_abs_rgb"), fxr.get("frac_rays_within_1e-4")
            cfgd["fp32_mlp_tflops"], cfgd["fp32_frac_of_fp32_mfma_peak"] = r.get("achieved_tflops_algorithmic"), r.get("frac_of_pipe_peak")
            other_acc = "f16x2" if args.mlp_precision == "f16x2s" else "f16x2s"
            r = short(f"mlp_precision_{other_acc} (split-half operands with {'two accumulators, residual halves scaled by 2^11: the default of rounds 1-3' if other_acc == 'f16x2' else 'one accumulator'})",
                      precision=other_acc, keep_frame0=True)
            if other_acc == "f16x2":
                cfgd["two_accumulator_f16x2_rays_per_s"] = r.get("value")
                cfgd["two_accumulator_f16x2_max_abs_rgb_vs_reference"] = r.get("vs_reference_fixture", {}).get("max_abs_rgb")
            r = short("mlp_precision_" + args.mlp_precision.split("+")[0] + "+f16col (split-half geometry network, whose error the s = 400 sigmoid amplifies, + ONE f16 product in the "
                      "colour network, whose error is damped by the sigmoid's slope <= 1/4: error-quantified, not the default)",
                      precision=args.mlp_precision.split("+")[0] + "+f16col", keep_frame0=True)
            fxr = r.get("vs_reference_fixture", {})
            cfgd["f16col_rays_per_s"], cfgd["f16col_ms_per_frame"] = r.get("value"), r.get("ms_per_frame")
            cfgd["f16col_max_abs_rgb_vs_reference"], cfgd["f16col_frac_rays_within_1e-4"] = fxr.get("max_abs_rgb"), fxr.get("frac_rays_within_1e-4")
            cfgd["f16col_max_abs_rgb_vs_headline_frame"] = r.get("max_abs_rgb_vs_headline_frame")
            r = short("mlp_precision_f16 (ONE f16 MFMA per product, 11-bit operands: the 'bf16 MLP'-class mode of BASELINE configs[1]; misses the 1e-4 bound, never a default)",
                      precision="f16", keep_frame0=True)
            cfgd["f16_single_rays_per_s"], cfgd["f16_single_ms_per_frame"] = r.get("value"), r.get("ms_per_frame")
            fxr = r.get("vs_reference_fixture", {})
            cfgd["f16_single_max_abs_rgb_vs_reference"], cfgd["f16_single_psnr_db_vs_reference"] = fxr.get("max_abs_rgb"), fxr.get("psnr_db")
            cfgd["f16_single_frac_rays_within_1e-4"] = fxr.get("frac_rays_within_1e-4")
            short("calc_normal_false", normals=False)
            short("two_half_frame_chunks_on_two_streams (rayschunk = half a frame: the low-occupancy per-ray kernels of one chunk run beside the other chunk's "
                  "kernels; identical pixels; the per-kernel event times of this run overlap, so the roofline figures are taken from the one-stream headline run)",
                  chunk=(n_rays + 1) // 2, keep_frame0=True)
            from neumesh_amd import renderer as _rmod
            r = short(f"library_default_chunks (the caller passes render.py's rayschunk = 4096, the library cuts the call into equal chunks of at most {_rmod.DEFAULT_RAYSCHUNK} rays "
                      f"-- 20.6 GB of workspace per lane, the same 41 GB in all as the one-call frame --, alternating between {_rmod.DEFAULT_LANES} streams; identical pixels)", library_policy=True, keep_frame0=True)
            r65 = short("chunks_of_65536_rays (the round-4 library default; identical pixels)", chunk=65536)
            cfgd["rayschunk_65536_ms_per_frame"] = r65.get("ms_per_frame")
            cfgd["default_rayschunk_ms_per_frame"] = r.get("ms_per_frame")
            # what a caller of volume_render gets without naming a chunk (render.py:211-218 passes 4096: a lower bound here), beside the one-call headline
            out["value_library_default_chunks"], out["ms_per_frame_library_default_chunks"] = r.get("value"), r.get("ms_per_frame")
            out["value_is"] = ("`value` / `roofline`: the frame as ONE nm_render_rays call on one stream (every kernel launch alone on the chip: the per-kernel event times are "
                               "the kernels' own).  `value_library_default_chunks`: the same frame through the library's own chunking -- equal chunks of <= 327 680 rays "
                               "alternating between two streams, what render.py's renderer(...) call gets -- whose kernels overlap across the two streams (identical pixels)")
            short("weight_eps_1e-10 (mid-points of visibility weight < 1e-10 not evaluated: the one variant that is not bit-identical; "
                  "rgb / normals move by < 127e-10, depth / acc not at all)", weight_eps=1e-10, keep_frame0=True)
            short("config3_shape (64 samples/ray, white background)", samples=64, white=True)
            short("config4_shape (1600x1200 rays/frame in chunks of one 800x800 frame, 64+64 samples)", hw=(1200, 1600))
            try:   # the other scene (rounds 1-2 headline: default-init noise field), same kernels
                other = "noise" if args.scene == "surf" else "surf"
                _, model2 = build_scene(args.V, dev, scene=other)
                r = short(f"{other}_scene (same shape on the {'default-init noise field, s = 200: every ray opaque' if other == 'noise' else 'scene with a surface'})", mdl=model2)
                cfgd[f"{other}_scene_rays_per_s"], cfgd[f"{other}_scene_ms_per_frame"] = r.get("value"), r.get("ms_per_frame")
                del model2
            except Exception as ex:
                extra["other_scene"] = {"error": str(ex)}
            model.mlp_precision = args.mlp_precision
            extra.update(consumer_rows(mesh, model, dev, args.H, args.W))
            for k_, v_ in extra.items():
                if k_.startswith("train_step") and "ms_per_step" in v_:
                    cfgd["train_step_ms_512_rays"], cfgd["train_step_ms_512_rays_torch_autograd_field"] = v_["ms_per_step"], v_.get("ms_per_step_torch_autograd_field")
            try:   # the TRAINED field (round 6; after the consumer rows: a 37 MB checkpoint load in this process was followed by a 2.4x slower training-step row): same shape on tests/golden/trained_v140k.pt, with its parity against the imported reference's render
                   # of these very rays and whether the call tripped the fp16-range flag (it would have fallen back to the fp32 kernels)
                if args.V == 140_000 and args.scene != "trained":
                    _, model3 = build_scene(args.V, dev, scene="trained")
                    r = short("trained_scene (tests/golden/trained_v140k.pt: 20 000 iterations of the reference's training recipe on an analytic scene, s = 1000; "
                              "loaded as render.py:287-288 does)", mdl=model3, keep_frame0=True)
                    cfgd["trained_scene_rays_per_s"], cfgd["trained_scene_ms_per_frame"] = r.get("value"), r.get("ms_per_frame")
                    cfgd["trained_scene_mlp_precision_after_render"] = model3.mlp_precision
                    img3 = getattr(run, "last_rgb0", None)
                    pb = parity_blocks(img3, args.H, args.W, args.V, None, None, "trained", model=model3).get("parity_vs_reference")
                    if pb and name_of_last_short[0] in extra:
                        extra[name_of_last_short[0]]["parity_vs_reference"] = pb
                        cfgd["trained_scene_parity_max_abs_rgb_vs_reference"] = pb["max_abs_rgb"]
                        cfgd["trained_scene_parity_frac_rays_within_1e-4"] = pb["frac_rays_within_1e-4"]
                        cfgd["trained_scene_parity_on_reference_depths_max_abs_rgb"] = pb.get("on_reference_depths", {}).get("max_abs_rgb")
                        cfgd["trained_scene_reference_self_1ulp_frac_rays_within_1e-4"] = pb["reference_self_sensitivity_1ulp"]["frac_rays_within_1e-4"]
                    del model3
            except Exception as ex:
                extra["trained_scene"] = {"error": str(ex)[-300:]}
        if world == 1 and args.cpu_rays > 0:
            try:
                r0 = (rays0[0].cpu().numpy(), rays0[1].cpu().numpy())
                base = None
                try:
                    from oracle.refimport import harness
                    if harness.reference_available():   # (the build container; the GPU box has no reference tree)
                        base, orgb, sel = reference_baseline(mesh, model, args.H, args.W, min(args.cpu_rays, 512), r0, samples=args.samples,
                                                             white_bkgd=args.white_bkgd, calc_normal=not args.no_normals)
                except Exception:
                    base = None
                if base is None:
                    base, orgb, sel = cpu_baseline(mesh, model, args.H, args.W, args.cpu_rays, r0, samples=args.samples,
                                                   white_bkgd=args.white_bkgd, calc_normal=not args.no_normals)
                out["cpu_baseline"] = base
                out["speedup_vs_cpu_baseline"] = value / base["value"]
                out.update(parity_blocks(rgb0, args.H, args.W, args.V, orgb, sel, args.scene, model=model if (args.samples == 128 and not args.white_bkgd and not args.no_extras) else None))
            except Exception as e:  # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "rays/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        elif world == 1:
            out.update(parity_blocks(rgb0, args.H, args.W, args.V, None, None, args.scene, model=model if (args.samples == 128 and not args.white_bkgd and not args.no_extras) else None))
        if "parity_vs_reference" in out:
            pr = out["parity_vs_reference"]
            cfgd["parity_max_abs_rgb_vs_reference"], cfgd["parity_median_abs_rgb_vs_reference"] = pr["max_abs_rgb"], pr["median_abs_rgb"]
            cfgd["parity_frac_rays_within_1e-4"] = pr["frac_rays_within_1e-4"]
            cfgd["parity_on_reference_depths_max_abs_rgb"] = pr.get("on_reference_depths", {}).get("max_abs_rgb")
            cfgd["reference_self_1ulp_frac_rays_within_1e-4"] = pr["reference_self_sensitivity_1ulp"]["frac_rays_within_1e-4"]
        if world == 1 and not args.no_extras and extra:
            try:   # BASELINE config 5 (HBM-stress of the K-NN + gather kernel), 2 steps
                del model
                from neumesh_amd.renderer import release_workspaces
                release_workspaces()
                torch.cuda.empty_cache()
                s5 = stress5_run(args, dev, 1, 0, 2, 1)
                extra["config5_stress (V=1M, 256-d table, 4096x4096 queries/step)"] = {
                    "value": s5["value"], "unit": s5["unit"], "ms_per_step": s5["ms_per_step"], "steps": 2, "roofline": s5["roofline"]}
                cfgd["config5_queries_per_s"] = s5["value"]
                cfgd["config5_frac_algorithmic_of_hbm_peak"] = s5["roofline"]["frac"]
                mh = s5["roofline"].get("measured_hbm_GBs")
                cfgd["config5_frac_measured_hbm_of_peak"] = (mh / PEAK_HBM_GBS) if mh else None
            except Exception as ex:
                extra["config5_stress (V=1M, 256-d table, 4096x4096 queries/step)"] = {"error": str(ex)}
        timer.cancel()
        emit()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
