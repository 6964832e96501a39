"""Volume renderer -- host-side mirror of the reference's ``models/renderer.py``.

``volume_render`` keeps the reference's signature (models/renderer.py:105-135, unknown kwargs
swallowed by ``**dummy_kwargs``) and return value ``(rgb, depth, ret_dict)`` (:368);
``SingleRenderer`` is the same thin nn.Module (:371-377).  For a NeuMesh field under
``torch.no_grad()`` with the deterministic sampler (what render.py uses: perturb=False) each ray
chunk is ONE call of ``nm_render_rays``: ray set-up, 256-probe near/far tightening, coarse
samples, 4x hierarchical up-sampling, SDF/nabla/colour queries and compositing all run as HIP
kernels on the current stream, with no per-stage tensors materialised in Python.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import _lib
from .neumesh import NeuMesh


def cdf_Phi_s(x, s):
    return torch.sigmoid(x * s)


def sdf_to_alpha(sdf, s):
    """models/renderer.py:17-24."""
    cdf = cdf_Phi_s(sdf, s)
    alpha = (cdf[..., :-1] - cdf[..., 1:]) / (cdf[..., :-1] + 1e-10)
    return cdf, torch.clamp_min(alpha, 0)


def alpha_to_w(alpha):
    """models/renderer.py:49-63."""
    ones = torch.ones([*alpha.shape[:-1], 1], device=alpha.device)
    return alpha * torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-10], dim=-1), dim=-1)[..., :-1]


# Evaluation-strategy switches of nm_render_rays (nm_render_cfg.flags / tuning fields).  The library
# itself reads no environment; this host layer maps the NEUMESH_* variables onto the struct so that a
# measurement script can flip them without touching the caller (none of them changes a result bit).
_ENV_FLAGS = (("NEUMESH_FULL_PROBES", _lib.RENDER_FULL_PROBES), ("NEUMESH_NO_ZERO_SKIP", _lib.RENDER_NO_ZERO_SKIP),
              ("NEUMESH_NO_RAY_SORT", _lib.RENDER_NO_RAY_SORT), ("NEUMESH_NO_MID_ORDER", _lib.RENDER_NO_MID_ORDER),
              ("NEUMESH_EAGER_NABLAS", _lib.RENDER_EAGER_NABLAS))
_ENV_TUNING = (("NEUMESH_CHAIN_TILES", "chain_tiles"), ("NEUMESH_FINE_GROUP", "fine_group_rays"), ("NEUMESH_MID_GROUP", "mid_group_rays"),
               ("NEUMESH_MID_PASSES", "mid_passes"))


def make_render_cfg(obj_bounding_radius=1.0, N_samples=64, N_importance=64, N_upsample_iters=4, bounded_near_far=True,
                    calc_normal=False, white_bkgd=False, near_bypass=None, far_bypass=None, flags=None,
                    weight_eps=None, **tuning) -> _lib.RenderCfg:
    """nm_render_cfg for volume_render's arguments.  flags: NM_RENDER_* bits (None = take them from the
    NEUMESH_* environment variables); tuning: chain_tiles / fine_group_rays / mid_group_rays / mid_passes (0 = default);
    weight_eps: visibility weights below it count as 0 (None = NEUMESH_WEIGHT_EPS, else 0 = exact; the only
    setting that changes pixels: by less than (N-1) * weight_eps)."""
    c = _lib.RenderCfg()
    c.obj_bounding_radius = float(obj_bounding_radius)
    c.N_samples, c.N_importance, c.N_upsample_iters = int(N_samples), int(N_importance), int(N_upsample_iters)
    c.bounded_near_far, c.calc_normal, c.white_bkgd = int(bool(bounded_near_far)), int(bool(calc_normal)), int(bool(white_bkgd))
    c.probe_grid, c.probe_thresh = 256, 0.1  # compute_bounded_near_far defaults (renderer.py:72-73)
    c.near_bypass = -1.0 if near_bypass is None else float(near_bypass)
    c.far_bypass = -1.0 if far_bypass is None else float(far_bypass)
    if flags is None:
        flags = 0
        for env, bit in _ENV_FLAGS:
            if os.environ.get(env):
                flags |= bit
    c.flags = int(flags)
    if weight_eps is None:
        try:
            weight_eps = float(os.environ.get("NEUMESH_WEIGHT_EPS", "0"))
        except ValueError:
            weight_eps = 0.0
    c.weight_eps = max(0.0, float(weight_eps))
    for env, field in _ENV_TUNING:
        v = tuning.get(field)
        if v is None:
            try:
                v = int(os.environ.get(env, "0"))
            except ValueError:
                v = 0
        setattr(c, field, max(0, int(v)))
    return c


class _Workspace:
    """Caller-owned scratch for nm_render_rays, reused across chunks / frames (one per chunk lane)."""

    def __init__(self):
        self.buf = None
        self.stream = None

    def get(self, nbytes: int, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = None
            self.buf = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        return self.buf

    def trim(self, keep_bytes: int):
        """Give a workspace larger than `keep_bytes` back to torch's allocator (the block stays cached there and can serve other
        allocations -- a training step after a validation render -- instead of sitting pinned in the pool)."""
        if self.buf is not None and self.buf.numel() > keep_bytes:
            self.buf = None

    def side_stream(self, device):
        if self.stream is None or self.stream.device != device:
            self.stream = torch.cuda.Stream(device=device)
        return self.stream


# Ray chunks are independent, and a chunk's launch sequence has under-filled stretches (the per-ray
# kernels occupy 2 waves per CU, every launch ends in a tail): when a call needs several chunks,
# consecutive chunks go to alternating HIP streams, each with its own workspace, so that one chunk's
# tails are filled by the other's kernels (800x800 frame on one MI355X: 65536-ray chunks 1029 ->
# 909 ms, 327680-ray chunks 951 -> 895 ms; the whole frame as ONE chunk, 925 ms, is single-stream).
# NEUMESH_RENDER_STREAMS=1 restores the single-stream order.
# Workspaces and side streams belong to one (device, caller stream) pair: calls issued on the same
# stream are ordered by it (fork/join below), calls on different streams or devices never share scratch.
# rays per nm_render_rays call unless NEUMESH_RAYSCHUNK says otherwise (_fused_chunk).  Round 5, tools/overlap_sweep.py and the bench line's rows
# (800x800 frame, same process): chunks of 65 536 rays on two lanes 325-352 ms (and no better on 3 / 4 / 6 lanes), four chunks of 160 000: 310.7-327.9,
# one call: 311.5-329.0, TWO chunks of 320 000 on two lanes: 304.3-319.7 -- always the fastest: a chunk is ~26 launches plus their tails, and one chunk's
# tails run under the other's kernels.  320 Ki rays = 20.6 GB of workspace per lane (63 KB per ray), 41 GB pooled for two lanes of the 288 GB;
# the chunk is halved while the lanes' workspaces would not fit a QUARTER of the free device memory, so a shared or nearly full GPU gets small chunks.
DEFAULT_RAYSCHUNK = 320 * 1024
WS_KEEP_BYTES = int(float(os.environ.get("NEUMESH_WS_KEEP_GB", "24")) * (1 << 30))   # pooled workspaces above this are returned after the call
MAX_LANES = 8               # chunk lanes (streams with a workspace each) a call may use
_POOLS = OrderedDict()      # (device, caller stream) -> its lanes' workspaces; least recently used first
_POOLS_LOCK = threading.Lock()
_POOLS_MAX = 4              # pools kept (each holds up to one full-chunk workspace per lane in use): callers on many short-lived streams
                            # (nn.DataParallel worker threads, per-request streams) must not accumulate them


def _lanes_for(device, stream_handle: int):
    key = (device.index if device.index is not None else torch.cuda.current_device(), int(stream_handle or 0))
    with _POOLS_LOCK:
        pool = _POOLS.pop(key, None)
        if pool is None:
            pool = [_Workspace() for _ in range(MAX_LANES)]
        _POOLS[key] = pool                      # most recently used last
        while len(_POOLS) > _POOLS_MAX:
            _POOLS.popitem(last=False)          # (its tensors are freed once the evicted call's own references go)
    return pool


def release_workspaces():
    """Drop every cached render workspace (they are sized for the largest chunk seen: 63 KB per ray at 32 + 32-d codes)."""
    with _POOLS_LOCK:
        _POOLS.clear()


DEFAULT_LANES = 2            # (more lanes measured no better, see DEFAULT_RAYSCHUNK)


def _env_int(name: str, default: int) -> int:
    try:
        return int(os.environ.get(name, "") or default)
    except ValueError:
        return default


def _n_lanes() -> int:
    """Ray chunks of one call in flight at a time (NEUMESH_RENDER_STREAMS): each on its own stream with its own workspace."""
    return max(1, min(MAX_LANES, _env_int("NEUMESH_RENDER_STREAMS", DEFAULT_LANES)))


def fusable_edit_model(model) -> bool:
    """A TextureEditableNeuMesh whose texture blend nm_render_rays can do itself (nm_render_cfg.n_edit): plain NeuMesh main and
    reference models with the main model's colour configuration, at most 4 references (their rotations, if any, go along)."""
    from .editing import TextureEditableNeuMesh
    if not isinstance(model, TextureEditableNeuMesh) or not isinstance(model.main_model, NeuMesh):
        return False
    refs = list(model.ref_models)
    m = model.main_model
    return (1 <= len(refs) <= 4 and all(isinstance(r, NeuMesh) for r in refs) and (model.rot_s_m is None or model.rot_s_m.shape[0] == len(refs)) and
            all(r.color_features.shape[1] == m.color_features.shape[1] and r.enable_nablas_input == m.enable_nablas_input for r in refs) and
            model.main_editing_masks.shape[0] == len(refs))


def render_rays_fused(model, rays_o, rays_d, cfg: _lib.RenderCfg, rayschunk: int, detailed: bool = False,
                      tables=None, progress=None, perturb: bool = False):
    """rays_o / rays_d: [R,3] device tensors.  Returns dict of [R,...] tensors.  model: a NeuMesh, or a
    TextureEditableNeuMesh that fusable_edit_model() accepts (its blend then runs inside nm_render_rays).
    perturb: importance samples by sample_pdf(det=False) -- one torch.rand block [iterations, rays, new samples] per ray chunk, handed
    to the kernels through nm_render_cfg.u_rand (ABI v9)."""
    main, keep = model, []
    mine = _lib.RenderCfg()   # (the edit_* pointers below are only valid during this call: the caller's struct stays untouched)
    C.memmove(C.byref(mine), C.byref(cfg), C.sizeof(_lib.RenderCfg))
    cfg = mine
    if not isinstance(model, NeuMesh):
        main = model.main_model
        refs = list(model.ref_models)
        masks = model.main_editing_masks.to(torch.uint8).contiguous()
        feats = model.main_editing_colorfeats.detach().float().contiguous()
        cfg.n_edit = len(refs)
        for i, r in enumerate(refs):
            cfg.edit_field[i] = getattr(r.field_handle(), "value", r.field_handle())
            cfg.edit_mask[i] = masks[i].data_ptr()
        cfg.edit_color_features = feats.data_ptr()
        rots = None if model.rot_s_m is None else model.rot_s_m.detach().float().cpu().numpy()
        for i in range(len(refs)):
            cfg.edit_use_rot[i] = 0 if rots is None else 1
            if rots is not None:
                for j in range(9):
                    cfg.edit_rot[i][j] = float(rots[i].reshape(-1)[j])
        keep = [masks, feats, refs]
    else:
        cfg.n_edit = 0
    models = [main] + (keep[2] if keep else [])
    u_blocks = {} if (perturb and cfg.N_importance > 0) else None   # (chunk start -> its numbers: drawn once, so that the fp32 re-run
                                                                     #  below, should it happen, places the same samples)
    plan = {}                                                        # (and the chunk size chosen once: u_blocks is shaped by it)
    # fp16-range flag of the split-half kernels (sticky on the device; depends on weights AND inputs).  No hidden sync (the reference
    # returns unsynchronised tensors, models/renderer.py:353-368): the flag is read WITH a stream sync -- and the call repeated by the fp32
    # kernels before it returns -- only on the first fused call on a weight set, under detailed_output, or with NEUMESH_EAGER_RANGE_CHECK=1.
    # Every other call posts an asynchronous read (nm_field_overflow_post) that the next entry below, or model.synchronize_fp16_range(),
    # evaluates: an overflow then costs a warning naming the earlier call(s) as invalid, and the model runs fp32 from there on.
    for m in models:
        if m._range_pending and not m.poll_fp16_range():
            for i, r in enumerate(keep[2] if keep else []):     # (now fp32: new handles)
                cfg.edit_field[i] = getattr(r.field_handle(), "value", r.field_handle())
    eager_env = os.environ.get("NEUMESH_EAGER_RANGE_CHECK", "")
    eager = eager_env == "1" or (eager_env != "0" and (detailed or not all(m._range_checked for m in models)))
    out = _render_rays_fused(main, rays_o, rays_d, cfg, rayschunk, detailed, tables, progress, u_blocks, plan)
    if not eager:
        for m in models:
            m.post_fp16_range_check()
    elif not all([m.check_fp16_range() for m in models]):   # a value left the fp16 range during this call: fp32 kernels, once more
        for i, r in enumerate(keep[2] if keep else []):
            cfg.edit_field[i] = getattr(r.field_handle(), "value", r.field_handle())
        out = _render_rays_fused(main, rays_o, rays_d, cfg, rayschunk, detailed, tables, progress, u_blocks, plan)
    del keep
    return out


def _fused_chunk(lib, cfg, R: int, rayschunk: int, dev, extra_per_ray: int = 0, held_bytes: int = 0) -> int:
    """Rays per nm_render_rays call.  The reference's `rayschunk` (render.py passes 4096) bounds ITS memory; here every chunk is ~26 kernel
    launches whose cost is latency, not work, below ~10^5 rays (800x800 frame: 806 ms in chunks of 4096 rays, 362 ms at 65 536, 354 ms in
    one call) and the pixels do not depend on the chunking (bit-identical, tested), so the caller's value is only a LOWER bound: the call is
    cut into equal chunks of at most NEUMESH_RAYSCHUNK rays.  The default is DEFAULT_RAYSCHUNK = 327 680 (round 5: an 800x800 frame as two chunks
    on two streams is 2-3 % FASTER than one call at the same 41 GB of workspace; 65 536, the round-4 default, measured 3-5 % slower than one call); it is
    halved while the lanes' workspaces plus what the call itself allocates per ray (`extra_per_ray`: the detailed-output tensors) would take more than a
    quarter of the free device memory, so a validation render during training or a shared GPU keeps its memory.  NEUMESH_RAYSCHUNK=0 honours the caller's value
    exactly; a larger value (bench.py: the whole frame) trades memory for the last 2 %."""
    want = max(1, min(int(rayschunk), R))
    own = int(os.environ.get("NEUMESH_RAYSCHUNK") or DEFAULT_RAYSCHUNK)
    if own <= want:
        return want
    chunk = min(R, own)
    try:
        free = torch.cuda.mem_get_info(dev)[0]
    except Exception:
        return want
    free += int(held_bytes)                   # workspaces this caller's pool already holds serve the call: they are not somebody else's memory
    free -= int(extra_per_ray) * R            # tensors of the whole call (allocated before the first chunk runs)
    while chunk > want:
        need = int(lib.nm_render_workspace_bytes(C.byref(cfg), chunk)) * (1 if chunk >= R else min(_n_lanes(), -(-R // chunk)))
        if 0 <= need <= free // (2 if os.environ.get("NEUMESH_RAYSCHUNK") else 4):   # (a size the user named: half of the free memory)
            break
        chunk = max(want, chunk // 2)
    if os.environ.get("NEUMESH_RAYSCHUNK"):
        return chunk                              # a size the user named is taken as named
    # the built-in default: equal chunks -- 640 000 rays are two chunks of 320 000, not one of 327 680 and one of 312 320 (the lanes finish together)
    balanced = -(-R // -(-R // chunk))
    return balanced if balanced >= want else chunk


def _render_rays_fused(model: NeuMesh, rays_o, rays_d, cfg, rayschunk, detailed, tables, progress, u_blocks=None, plan=None):
    lib = _lib.load()
    dev = rays_o.device
    if dev.type != "cuda":
        raise _lib.NeuMeshHipError("rays must be on a HIP device (no CPU fallback)")
    rays_o = rays_o.detach().float().reshape(-1, 3).contiguous()
    rays_d = rays_d.detach().float().reshape(-1, 3).contiguous()
    R = rays_o.shape[0]
    N = cfg.N_samples + cfg.N_importance
    out = OrderedDict(rgb=torch.empty((R, 3), device=dev), depth_volume=torch.empty((R,), device=dev),
                      mask_volume=torch.empty((R,), device=dev))
    if cfg.calc_normal:
        out["normals_volume"] = torch.empty((R, 3), device=dev)
    dbg_t = {}
    if detailed:
        dbg_t = dict(d_all=torch.empty((R, N), device=dev), sdf_all=torch.empty((R, N), device=dev),
                     radiance=torch.empty((R, N - 1, 3), device=dev), near_far=torch.empty((R, 2), device=dev))
        if cfg.calc_normal:
            dbg_t["nablas_all"] = torch.empty((R, N, 3), device=dev)
    cfg.code_dims = int(model._cfg["geometry_dim"]) | (int(model._cfg["color_dim"]) << 16)   # K-NN records of the workspace sized for this field
    # detailed output: ~12 more [R, N]-sized tensors are derived from the debug arrays after the last chunk
    with torch.cuda.device(dev):
        pool = _lanes_for(dev, torch.cuda.current_stream(dev).cuda_stream)
    if plan is not None and "chunk" in plan:     # the fp32 re-run of a call: the SAME chunks (u_blocks is keyed and shaped by them)
        chunk = plan["chunk"]
    else:
        held = sum(int(lane.buf.numel()) for lane in pool if lane.buf is not None and lane.buf.device == dev)   # (this pool's own workspaces are
        chunk = _fused_chunk(lib, cfg, R, rayschunk, dev, extra_per_ray=4 * N * 12 if detailed else 0, held_bytes=held)   #  not "used" memory: ADVICE r5)
        if plan is not None:
            plan["chunk"] = chunk
    ws_bytes = int(lib.nm_render_workspace_bytes(C.byref(cfg), chunk))
    if ws_bytes < 0:
        _lib.check(1, "nm_render_workspace_bytes")
    starts = list(range(0, R, chunk))
    if u_blocks is not None:   # sample_pdf(det=False): drawn on the caller's stream BEFORE the chunk streams fork from it
        n_new = cfg.N_importance // cfg.N_upsample_iters
        for i in starts:
            if i not in u_blocks:
                u_blocks[i] = torch.rand((cfg.N_upsample_iters, min(chunk, R - i), n_new), dtype=torch.float32, device=dev)
    field, grid = model.field_handle(), model.grid_for(dev).grid.handle
    t, keep = tables if tables is not None else model.field_tables()
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        lanes = pool[:min(_n_lanes(), len(starts))]
        wss = [lane.get(ws_bytes, dev) for lane in lanes]
        if len(lanes) > 1:  # fork: the side streams start after everything already queued on the caller's stream
            side = [lane.side_stream(dev) for lane in lanes]
            for st in side:
                st.wait_stream(main)
            streams = [C.c_void_p(st.cuda_stream) for st in side]
        else:
            side, streams = [], [_lib.current_stream(dev)]
        for ci, i in enumerate(starts if progress is None else progress(starts)):
            n = min(chunk, R - i)
            ws, stream = wss[ci % len(lanes)], streams[ci % len(lanes)]
            dbg = None
            if detailed:
                dbg = _lib.RenderDebug()
                dbg.near_far = dbg_t["near_far"][i:].data_ptr()
                dbg.d_all = dbg_t["d_all"][i:].data_ptr()
                dbg.sdf_all = dbg_t["sdf_all"][i:].data_ptr()
                dbg.radiance = dbg_t["radiance"][i:].data_ptr()
                dbg.nablas_all = dbg_t["nablas_all"][i:].data_ptr() if cfg.calc_normal else None
                dbg.sdf_coarse = None
            cfg.u_rand = u_blocks[i].data_ptr() if u_blocks is not None else None
            _lib.check(lib.nm_render_rays(
                field, grid, C.byref(t), _lib.ptr(rays_o[i:]), _lib.ptr(rays_d[i:]), n, C.byref(cfg),
                _lib.ptr(out["rgb"][i:]), _lib.ptr(out["depth_volume"][i:]), _lib.ptr(out["mask_volume"][i:]),
                _lib.ptr(out["normals_volume"][i:]) if cfg.calc_normal else None,
                C.byref(dbg) if dbg is not None else None, _lib.ptr(ws), stream), "nm_render_rays")
        for st in side:  # join
            main.wait_stream(st)
        wss = ws = None   # (drop this call's references before the pools decide what to keep)
        for lane in lanes:   # (after the join: the block is handed back in the caller's stream order)
            lane.trim(WS_KEEP_BYTES)
        for lane in pool[len(lanes):]:   # lanes this chunking did not use keep nothing pinned
            lane.trim(0)
    del keep
    if detailed:
        s = model.forward_s().detach()
        sdf, d_all = dbg_t["sdf_all"], dbg_t["d_all"]
        cdf, alpha = sdf_to_alpha(sdf, s)
        if cfg.calc_normal:
            out["implicit_nablas"] = dbg_t["nablas_all"]
        out["implicit_surface"] = sdf
        out["radiance"] = dbg_t["radiance"]
        out["alpha"] = alpha
        out["cdf"] = cdf
        out["visibility_weights"] = alpha_to_w(alpha)
        out["d_final"] = 0.5 * (d_all[..., 1:] + d_all[..., :-1])
        out["d_all"] = d_all              # extra (not in the reference's dict)
        out["near_far"] = dbg_t["near_far"]  # extra
    return out


class _RestoreGradMode:
    """The staged renderer switches autograd off for the sample placement of every chunk and back on
    for the differentiable tail; this puts the caller's mode back even if a stage raises."""

    def __enter__(self):
        self.mode = torch.is_grad_enabled()

    def __exit__(self, *exc):
        torch.set_grad_enabled(self.mode)
        return False


def _fused_sample(lib, model, ro, rd, cfg, perturb: bool, st):
    """Sample placement of one ray chunk by nm_render_rays(NM_RENDER_SAMPLE_ONLY): (sorted depths d [R,N], near/far [R,2])."""
    dev = ro.device
    R, N = ro.shape[0], cfg.N_samples + cfg.N_importance
    c = _lib.RenderCfg()
    C.memmove(C.byref(c), C.byref(cfg), C.sizeof(_lib.RenderCfg))
    c.flags = int(cfg.flags) | _lib.RENDER_SAMPLE_ONLY
    c.n_edit = 0
    c.code_dims = int(model._cfg["geometry_dim"]) | (int(model._cfg["color_dim"]) << 16)
    u = None
    if perturb and cfg.N_importance > 0:   # sample_pdf(det=False): one uniform number per new sample, per iteration (rend_util.py:300-302)
        u = torch.rand((cfg.N_upsample_iters, R, cfg.N_importance // cfg.N_upsample_iters), dtype=torch.float32, device=dev)
        c.u_rand = u.data_ptr()
    d = torch.empty((R, N), dtype=torch.float32, device=dev)
    nf = torch.empty((R, 2), dtype=torch.float32, device=dev)
    dbg = _lib.RenderDebug()
    dbg.d_all, dbg.near_far = d.data_ptr(), nf.data_ptr()
    ws_bytes = int(lib.nm_render_workspace_bytes(C.byref(c), R))
    if ws_bytes < 0:
        _lib.check(1, "nm_render_workspace_bytes")
    lane = _lanes_for(dev, torch.cuda.current_stream(dev).cuda_stream)[0]
    ws = lane.get(ws_bytes, dev)
    t, keep = model.field_tables()
    _lib.check(lib.nm_render_rays(model.field_handle(), model.grid_for(dev).grid.handle, C.byref(t), _lib.ptr(ro), _lib.ptr(rd), R, C.byref(c),
                                  None, None, None, None, C.byref(dbg), _lib.ptr(ws), st), "nm_render_rays(sample only)")
    del keep, u
    return d, nf


def render_rays_staged(model, rays_o, rays_d, cfg: _lib.RenderCfg, rayschunk: int, netchunk: int, detailed: bool = False,
                       progress=None, differentiable: bool = False, perturb: bool = False, trace=None,
                       samples_output: bool = False, random_color_direction: bool = False):
    """render_rayschunk (models/renderer.py:162-350) for ANY object that offers the field methods the
    reference's renderer calls -- compute_distance / forward_density_only / forward_with_nablas /
    forward / forward_s -- e.g. the editing tools' TextureEditableNeuMesh wrapper
    (editing/texture_neumesh/texture_neumesh.py:41-122).  Every per-ray stage runs as the same HIP
    kernel the fused path uses (C ABI nm_rays_*); between the stages the field is queried through the
    model's own methods, in chunks of `netchunk` points like train_util.batchify_query.

    differentiable=True is the training form (trainer.py:75-81): the sample placement runs exactly as
    above under torch.no_grad() (as in the reference, renderer.py:200-259), then the field is queried at
    the final points WITH autograd and alpha / weights / compositing are torch ops (renderer.py:264-333),
    so gradients reach every model parameter.  perturb=True draws the importance samples with
    sample_pdf(det=False) (torch.rand handed to nm_rays_upsample).

    samples_output (renderer.py:198,291-293,343-347): with `detailed`, the mid-points, their view directions, SDF
    and radiance are returned as "xyz" / "dirs" / "density" / "colors" (the distillation losses of
    models/trainer.py:211-221 read them).  random_color_direction (renderer.py:279-289): the colour branch is
    queried with random unit directions (torch.rand_like, normalised) instead of the rays' own.

    trace (diagnostics): a dict that receives, per ray chunk, the stage outputs a diverging ray can be
    followed through -- "near_far" [R,2], "sdf_coarse" [R,Ns], "d_iter" (list: sorted depths after each
    up-sampling iteration, renderer.py:255)."""
    lib = _lib.load()
    dev = rays_o.device
    rays_o = rays_o.detach().float().reshape(-1, 3).contiguous()
    rays_d = rays_d.detach().float().reshape(-1, 3).contiguous()
    Rall = rays_o.shape[0]
    Ns, Ni, iters = cfg.N_samples, cfg.N_importance, cfg.N_upsample_iters
    N = Ns + Ni
    # Processing order: rays sorted by the Morton code of their point of closest approach to the scene centre (what
    # nm_render_rays does on the device for the fused path), so that 16 consecutive rays are neighbours in space even when
    # the caller hands over random pixels (training: trainer.py draws N_rays random pixels per step); per-ray outputs go
    # back to the caller's order at the end.  (trace: diagnostics follow the caller's rays one by one -- no reordering.)
    # Fewer than 4096 rays per call are a training batch of random pixels (trainer.py draws 512 per step), not an image: neighbours
    # in the ray order are then ~35 pixels apart, farther than an octree leaf, so a 16-ray x 4-sample tile is NOT a compact packet
    # -- consecutive samples of ONE ray are (measured on a training step, K-NN kernel time: ray-major 7.1 ms, sorted + tiled 9.0).
    dense = Rall >= 4096
    ray_inv = None
    if dense and trace is None and not os.environ.get("NEUMESH_NO_RAY_SORT"):
        with torch.no_grad():
            dn = rays_d / torch.linalg.norm(rays_d, dim=-1, keepdim=True).clamp_min(1e-12)
            pc = rays_o - (rays_o * dn).sum(-1, keepdim=True) * dn
            qv = ((pc / max(float(cfg.obj_bounding_radius), 1e-6) * 0.5 + 0.5).clamp(0.0, 1.0) * 1023.0).to(torch.int64)
            for sh, msk in ((16, 0x30000FF), (8, 0x300F00F), (4, 0x30C30C3), (2, 0x9249249)):   # spread 10 bits to every third position
                qv = (qv | (qv << sh)) & msk
            code = qv[:, 0] | (qv[:, 1] << 1) | (qv[:, 2] << 2)
            ray_perm = torch.argsort(code, stable=True)
            ray_inv = torch.empty_like(ray_perm)
            ray_inv[ray_perm] = torch.arange(Rall, device=dev)
        rays_o, rays_d = rays_o[ray_perm].contiguous(), rays_d[ray_perm].contiguous()

    tile_perms = {}

    def tile_perm(R, P):
        """Order in which the points of an [R,P] block are handed to the model's point-wise methods: tiles of 16 adjacent
        rays x 4 consecutive samples (64 consecutive points = one compact packet for the wave-cooperative K-NN search --
        in ray-major order a wave would get 64 samples strung along ONE ray and fall back to lane-private traversals),
        then whatever does not fill a tile.  Any order gives the same per-point results."""
        key = (R, P)
        if not dense or os.environ.get("NEUMESH_NO_TILE_ORDER"):
            return None
        if key not in tile_perms:
            idx = torch.arange(R * P, device=dev).view(R, P)
            R16, P4 = R // 16 * 16, P // 4 * 4
            parts = []
            if R16 and P4:
                parts.append(idx[:R16, :P4].reshape(R16 // 16, 16, P4 // 4, 4).permute(0, 2, 1, 3).reshape(-1))
                parts.append(idx[:R16, P4:].reshape(-1))
                parts.append(idx[R16:, :].reshape(-1))
                perm = torch.cat(parts)
                inv = torch.empty_like(perm)
                inv[perm] = torch.arange(R * P, device=dev)
                tile_perms[key] = (perm, inv)
            else:
                tile_perms[key] = None
        return tile_perms[key]

    def query(fn, pts, *extra):  # pts [R,P,3] -> tuple of [R,P,...]
        flat = pts.reshape(-1, 3)
        ex = [e.reshape(-1, e.shape[-1]) for e in extra]
        tp = tile_perm(pts.shape[0], pts.shape[1])
        if tp is not None:
            flat = flat[tp[0]]
            ex = [e[tp[0]] for e in ex]
        outs = []
        for i in range(0, flat.shape[0], max(1, int(netchunk))):
            o = fn(flat[i:i + netchunk], *[e[i:i + netchunk] for e in ex])
            outs.append(o if isinstance(o, tuple) else (o,))
        cols = [torch.cat([o[k] for o in outs], 0) for k in range(len(outs[0]))]
        if tp is not None:
            cols = [c[tp[1]] for c in cols]
        return [c.reshape(pts.shape[0], pts.shape[1], *c.shape[1:]) for c in cols]

    # (fused sampler: below ~4096 rays a call is a training batch of random pixels, for which the fused kernels' 16-ray tiles and
    #  serial probe walk are the wrong shape -- measured on a 512-ray step: 27.1 ms against 16.4 ms with the stages below, the probe
    #  walk alone 8.3 ms -- so only dense calls take it; NEUMESH_FUSED_SAMPLER=1 / =0 force it on / off)
    fs_env = os.environ.get("NEUMESH_FUSED_SAMPLER")
    fused_sampler = (differentiable and trace is None and isinstance(model, NeuMesh) and model.fused_supported()
                     and (fs_env == "1" or (fs_env != "0" and min(Rall, max(1, int(rayschunk))) >= 4096)))
    chunks = []
    rng = range(0, Rall, max(1, int(rayschunk)))
    with torch.cuda.device(dev), _RestoreGradMode():
        st = _lib.current_stream(dev)
        for i in (rng if progress is None else progress(rng)):
            ro, rd = rays_o[i:i + rayschunk].contiguous(), rays_d[i:i + rayschunk].contiguous()
            R = ro.shape[0]
            f32 = dict(dtype=torch.float32, device=dev)
            grad_was = torch.is_grad_enabled()
            torch.set_grad_enabled(False)   # sample placement never carries gradients (renderer.py:200)
            dirn, nf0 = torch.empty((R, 3), **f32), torch.empty((R, 2), **f32)
            _lib.check(lib.nm_rays_setup(_lib.ptr(ro), _lib.ptr(rd), R, cfg.obj_bounding_radius, _lib.ptr(dirn), _lib.ptr(nf0), st), "nm_rays_setup")
            if fused_sampler:
                # Training step of a plain NeuMesh field: the whole sample placement (renderer.py:162-259, no_grad in the reference too)
                # is ONE C call -- nm_render_rays with NM_RENDER_SAMPLE_ONLY: first / last-hit probe walk instead of 256 searched probes
                # per ray, chained coarse tiles, slot records, depth-bucket lists, ~25 launches instead of ~60 kernels + torch glue --
                # returning the sorted depths (bit-identical to the stages below: the fused = staged tests) for the differentiable tail.
                d, nf = _fused_sample(lib, model, ro, rd, cfg, perturb, st)
                dmid = torch.zeros((R, N), **f32)
                dmid[:, :N - 1] = 0.5 * (d[:, 1:] + d[:, :-1])       # (renderer.py:266; the arithmetic of nm_rays_finalize)
                pts = torch.empty((R, N, 3), **f32)
                _lib.check(lib.nm_rays_points(_lib.ptr(ro), _lib.ptr(dirn), R, N, 1, None, _lib.ptr(d), N, 0, None, _lib.ptr(pts), st), "nm_rays_points")
                torch.set_grad_enabled(grad_was)
                chunks.append(_composite_autograd(model, query, cfg, ro, dirn, d, dmid, pts, detailed, nf, samples_output, random_color_direction))
                continue
            nf = nf0
            if cfg.bounded_near_far:
                G = cfg.probe_grid
                pts = torch.empty((R, G, 3), **f32)
                _lib.check(lib.nm_rays_points(_lib.ptr(ro), _lib.ptr(dirn), R, G, 2, _lib.ptr(nf0), None, G, 0, None, _lib.ptr(pts), st), "nm_rays_points")
                ds = model.compute_distance(pts)[0].reshape(R, G).float().contiguous()   # ONE call, like renderer.py:86
                nf = torch.empty((R, 2), **f32)
                _lib.check(lib.nm_rays_bounds(_lib.ptr(ds), R, G, cfg.probe_thresh, _lib.ptr(nf0), _lib.ptr(nf), st), "nm_rays_bounds")
            if cfg.near_bypass >= 0 or cfg.far_bypass >= 0:
                nf = nf.clone()
                if cfg.near_bypass >= 0:
                    nf[:, 0] = cfg.near_bypass
                if cfg.far_bypass >= 0:
                    nf[:, 1] = cfg.far_bypass
            d, sdf = torch.zeros((R, N), **f32), torch.zeros((R, N), **f32)
            pts = torch.empty((R, Ns, 3), **f32)
            _lib.check(lib.nm_rays_points(_lib.ptr(ro), _lib.ptr(dirn), R, Ns, 2, _lib.ptr(nf), None, N, 0, _lib.ptr(d), _lib.ptr(pts), st), "nm_rays_points")
            sdf[:, :Ns] = query(model.forward_density_only, pts)[0].reshape(R, Ns)
            if trace is not None:
                trace.setdefault("near_far", []).append(nf.clone())
                trace.setdefault("sdf_coarse", []).append(sdf[:, :Ns].clone())
                trace.setdefault("d_iter", []).append([])
            n, pending = Ns, 0
            if Ni > 0:
                n_new = Ni // iters
                for it in range(iters):
                    u = torch.rand((R, n_new), **f32) if perturb else None
                    _lib.check(lib.nm_rays_upsample(_lib.ptr(d), _lib.ptr(sdf), R, N, n, pending, it, n_new, _lib.ptr(u), st), "nm_rays_upsample")
                    pts = torch.empty((R, n_new, 3), **f32)
                    _lib.check(lib.nm_rays_points(_lib.ptr(ro), _lib.ptr(dirn), R, n_new, 1, None, _lib.ptr(d), N, n, None, _lib.ptr(pts), st), "nm_rays_points")
                    sdf[:, n:n + n_new] = query(model.forward_density_only, pts)[0].reshape(R, n_new)
                    n, pending = n + n_new, n_new
                    if trace is not None:   # (the merge of these n_new samples happens inside the next stage call)
                        trace["d_iter"][-1].append(torch.sort(d[:, :n], dim=-1)[0])
            dmid = torch.zeros((R, N), **f32)
            _lib.check(lib.nm_rays_finalize(_lib.ptr(d), _lib.ptr(sdf), R, N, n, pending, _lib.ptr(dmid), st), "nm_rays_finalize")
            pts = torch.empty((R, N, 3), **f32)
            _lib.check(lib.nm_rays_points(_lib.ptr(ro), _lib.ptr(dirn), R, N, 1, None, _lib.ptr(d), N, 0, None, _lib.ptr(pts), st), "nm_rays_points")
            torch.set_grad_enabled(grad_was)
            if differentiable:
                chunks.append(_composite_autograd(model, query, cfg, ro, dirn, d, dmid, pts, detailed, nf, samples_output, random_color_direction))
                continue
            ret = _staged_tail(lib, st, model, query, cfg, ro, dirn, d, dmid, pts, detailed, nf, samples_output, random_color_direction)
            chunks.append(ret)
    out = OrderedDict((k, torch.cat([c[k] for c in chunks], 0)) for k in chunks[0])
    if ray_inv is not None:
        out = OrderedDict((k, v[ray_inv]) for k, v in out.items())
    return out


def _staged_tail(lib, st, model, query, cfg, ro, dirn, d, dmid, pts, detailed, nf, samples_output=False, random_color_direction=False):
    """renderer.py:264-348 on given sorted depths d [R,N] (mid-point depths dmid, sample points pts): field + nablas at
    the samples, radiance at the mid-points through the model's own methods, nm_rays_composite."""
    R, N = d.shape
    dev = d.device
    f32 = dict(dtype=torch.float32, device=dev)
    nablas = None
    if cfg.calc_normal:
        s_all, nablas = query(model.forward_with_nablas, pts)
        nablas = nablas.float().contiguous()
    else:
        s_all = query(model.forward_density_only, pts)[0]
    sdf = s_all.reshape(R, N).float().contiguous()
    pm = torch.empty((R, N - 1, 3), **f32)
    _lib.check(lib.nm_rays_points(_lib.ptr(ro), _lib.ptr(dirn), R, N - 1, 1, None, _lib.ptr(dmid), N, 0, None, _lib.ptr(pm), st), "nm_rays_points")
    view = _mid_directions(dirn, pm, random_color_direction)
    sdf_mid, radiance = query(lambda x, v: model.forward(x, v)[:2], pm, view)
    radiance = radiance.float().contiguous()
    rgb, depth, acc = torch.empty((R, 3), **f32), torch.empty((R,), **f32), torch.empty((R,), **f32)
    normals = torch.empty((R, 3), **f32) if cfg.calc_normal else None
    s_val = float(model.forward_s())
    _lib.check(lib.nm_rays_composite(_lib.ptr(sdf), _lib.ptr(d), R, N, N, s_val, _lib.ptr(radiance), _lib.ptr(nablas), cfg.white_bkgd,
                                     _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(acc), _lib.ptr(normals), st), "nm_rays_composite")
    ret = OrderedDict(rgb=rgb, depth_volume=depth, mask_volume=acc)
    if cfg.calc_normal:
        ret["normals_volume"] = normals
    if detailed:
        cdf, alpha = sdf_to_alpha(sdf, s_val)
        if cfg.calc_normal:
            ret["implicit_nablas"] = nablas
        ret.update(implicit_surface=sdf, radiance=radiance, alpha=alpha, cdf=cdf, visibility_weights=alpha_to_w(alpha),
                   d_final=0.5 * (d[:, 1:] + d[:, :-1]), d_all=d, near_far=nf)
        if samples_output:
            ret.update(xyz=pm, dirs=dirn[:, None, :].expand(R, N - 1, 3), density=sdf_mid, colors=radiance)
    return ret


def render_at_depths(model, rays_o, rays_d, d_all, cfg: _lib.RenderCfg, netchunk: int = 1 << 20, detailed: bool = False):
    """The part of render_rayschunk after the sample placement (models/renderer.py:264-333) on GIVEN sorted sample
    depths d_all [R,N] (N = cfg.N_samples + cfg.N_importance): SDF (+ nablas) at the samples, radiance at the mid-points,
    alpha / visibility weights / compositing on the HIP kernels.  With another implementation's depths this compares
    everything behind the sampler ray by ray, without the sampler's sensitivity to the last bit of an SDF value."""
    lib = _lib.load()
    dev = rays_o.device
    if dev.type != "cuda":
        raise _lib.NeuMeshHipError("rays must be on a HIP device (no CPU fallback)")
    ro = rays_o.detach().float().reshape(-1, 3).contiguous()
    rd = rays_d.detach().float().reshape(-1, 3).contiguous()
    d = d_all.detach().float().contiguous()
    R, N = d.shape
    if R != ro.shape[0] or N != cfg.N_samples + cfg.N_importance:
        raise ValueError(f"d_all must be [{ro.shape[0]}, {cfg.N_samples + cfg.N_importance}], got {tuple(d.shape)}")
    f32 = dict(dtype=torch.float32, device=dev)

    def query(fn, pts, *extra):
        flat = pts.reshape(-1, 3)
        ex = [e.reshape(-1, e.shape[-1]) for e in extra]
        outs = []
        for i in range(0, flat.shape[0], max(1, int(netchunk))):
            o = fn(flat[i:i + netchunk], *[e[i:i + netchunk] for e in ex])
            outs.append(o if isinstance(o, tuple) else (o,))
        cols = [torch.cat([o[k] for o in outs], 0) for k in range(len(outs[0]))]
        return [c.reshape(pts.shape[0], pts.shape[1], *c.shape[1:]) for c in cols]

    with torch.cuda.device(dev), torch.no_grad():
        st = _lib.current_stream(dev)
        dirn, nf0 = torch.empty((R, 3), **f32), torch.empty((R, 2), **f32)
        _lib.check(lib.nm_rays_setup(_lib.ptr(ro), _lib.ptr(rd), R, cfg.obj_bounding_radius, _lib.ptr(dirn), _lib.ptr(nf0), st), "nm_rays_setup")
        dmid = torch.zeros((R, N), **f32)
        dmid[:, :N - 1] = 0.5 * (d[:, 1:] + d[:, :-1])
        pts = torch.empty((R, N, 3), **f32)
        _lib.check(lib.nm_rays_points(_lib.ptr(ro), _lib.ptr(dirn), R, N, 1, None, _lib.ptr(d), N, 0, None, _lib.ptr(pts), st), "nm_rays_points")
        return _staged_tail(lib, st, model, query, cfg, ro, dirn, d, dmid, pts, detailed, nf0)


def _mid_directions(dirn, pts_mid, random_color_direction: bool):
    """View directions handed to the colour branch at the mid-points: the ray's own direction, or
    (renderer.py:279-289) random ones -- torch.rand_like in [0,1)^3, normalised, exactly as the reference draws them."""
    R, M = pts_mid.shape[0], pts_mid.shape[1]
    if not random_color_direction:
        return dirn[:, None, :].expand(R, M, 3).contiguous()
    rnd = torch.rand_like(pts_mid)
    return rnd / torch.linalg.norm(rnd, axis=-1, keepdims=True)


_SIDE_STREAMS = {}


def _side_stream(device):
    """One extra stream per device for the second field query of the differentiable tail (kept: creating a stream costs a driver call)."""
    key = str(device)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class _HipComposite(torch.autograd.Function):
    """renderer.py:264-333 (sdf_to_alpha, alpha_to_w, the weighted sums) as ONE kernel forward and ONE backward (C ABI
    nm_train_composite_forward / _backward) instead of ~100 small torch kernels each way.  Differentiable outputs: rgb, depth, acc,
    normals (cotangents flow to sdf, radiance, nablas and s); cdf / alpha / weights are returned for the detailed outputs without a
    graph (nothing of the reference's trainer differentiates through them; NEUMESH_COMPOSITE=torch keeps the torch-op form)."""

    @staticmethod
    def forward(ctx, sdf, radiance, nablas, s, dmid, white):
        lib = _lib.load()
        dev = sdf.device
        R, N = sdf.shape
        f32 = dict(dtype=torch.float32, device=dev)
        sdf_c, rad_c = sdf.detach().float().contiguous(), radiance.detach().float().contiguous()
        nab_c = None if nablas is None else nablas.detach().float().contiguous()
        s_c, dm = s.detach().float().reshape(1).contiguous(), dmid.detach().float().contiguous()
        rgb, depth, acc = torch.empty((R, 3), **f32), torch.empty((R,), **f32), torch.empty((R,), **f32)
        normals = torch.empty((R, 3), **f32) if nab_c is not None else None
        cdf, alpha = torch.empty((R, N), **f32), torch.empty((R, N - 1), **f32)
        w, trans = torch.empty((R, N - 1), **f32), torch.empty((R, N - 1), **f32)
        with torch.cuda.device(dev):
            _lib.check(lib.nm_train_composite_forward(_lib.ptr(sdf_c), _lib.ptr(s_c), _lib.ptr(dm), dm.shape[1], _lib.ptr(rad_c), _lib.ptr(nab_c), R, N,
                                                      int(bool(white)), _lib.ptr(rgb), _lib.ptr(depth), _lib.ptr(acc), _lib.ptr(normals), _lib.ptr(cdf),
                                                      _lib.ptr(alpha), _lib.ptr(w), _lib.ptr(trans), _lib.current_stream(dev)), "nm_train_composite_forward")
        ctx.save_for_backward(sdf_c, rad_c, nab_c if nab_c is not None else sdf_c.new_zeros(0), s_c, dm, cdf, alpha, w, trans, acc, depth)
        ctx.has_nablas, ctx.white, ctx.s_shape = nab_c is not None, bool(white), tuple(s.shape)
        ctx.mark_non_differentiable(cdf, alpha, w)
        if normals is None:
            normals = rgb.new_zeros((R, 3))
            ctx.mark_non_differentiable(normals)
        return rgb, depth, acc, normals, cdf, alpha, w

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_acc, g_normals, *_unused):
        lib = _lib.load()
        sdf, rad, nab, s, dm, cdf, alpha, w, trans, acc, depth = ctx.saved_tensors
        nab = nab if ctx.has_nablas else None
        dev = sdf.device
        R, N = sdf.shape

        def c(g):
            return None if g is None else g.detach().float().contiguous()
        g_rgb, g_depth, g_acc, g_normals = c(g_rgb), c(g_depth), c(g_acc), (c(g_normals) if ctx.has_nablas else None)
        g_sdf = torch.empty_like(sdf)
        g_rad = torch.empty_like(rad) if ctx.needs_input_grad[1] else None
        g_nab = torch.empty_like(nab) if (nab is not None and ctx.needs_input_grad[2]) else None
        g_s = torch.zeros((1,), dtype=torch.float32, device=dev) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(dev):
            _lib.check(lib.nm_train_composite_backward(_lib.ptr(sdf), _lib.ptr(s), _lib.ptr(dm), dm.shape[1], _lib.ptr(rad), _lib.ptr(nab), R, N,
                                                       int(ctx.white), _lib.ptr(cdf), _lib.ptr(alpha), _lib.ptr(w), _lib.ptr(trans), _lib.ptr(acc),
                                                       _lib.ptr(depth), _lib.ptr(g_rgb), _lib.ptr(g_depth), _lib.ptr(g_acc), _lib.ptr(g_normals),
                                                       _lib.ptr(g_sdf), _lib.ptr(g_rad), _lib.ptr(g_nab), _lib.ptr(g_s), _lib.current_stream(dev)),
                       "nm_train_composite_backward")
        return g_sdf, g_rad, g_nab, (None if g_s is None else g_s.reshape(ctx.s_shape)), None, None


def _composite_autograd(model, query, cfg, ro, dirn, d, dmid, pts, detailed, nf, samples_output=False, random_color_direction=False):
    """renderer.py:264-348 on the (detached) sample depths d [R,N], differentiable: the field through the model's methods under autograd,
    alpha / weights / sums by _HipComposite (default) or torch ops (NEUMESH_COMPOSITE=torch)."""
    R, N = d.shape
    nablas = None
    pm = ro[:, None, :] + dmid[:, :N - 1, None] * dirn[:, None, :]
    view = _mid_directions(dirn, pm, random_color_direction)
    # The two field queries -- sdf + nablas at the samples, sdf + radiance at the mid-points -- do not depend on each other, and for a
    # training batch each is a few launch-latency-bound kernels (a 65 k-point K-NN launch lives ~1.1 ms for its slowest wave while
    # most of the chip idles): with NEUMESH_TRAIN_STREAMS=2 the mid-point query is issued on a second stream so that its kernels run
    # beside the sample query's (autograd runs each node's backward on the stream of its forward, so the backward passes overlap the
    # same way): 16.7 -> 15.9 ms per step in tools/train_profile.py.  Opt-in: the default bench run stalled once with it on (cause not
    # found within the round's GPU budget), and a stall is not worth a millisecond.
    side = _side_stream(pm.device) if (pm.is_cuda and pm.shape[0] * N <= (1 << 19) and os.environ.get("NEUMESH_TRAIN_STREAMS", "1") == "2") else None
    if side is not None:
        main = torch.cuda.current_stream(pm.device)
        side.wait_stream(main)
        pm.record_stream(side)        # allocated on the caller's stream, read by the side stream's kernels
        view.record_stream(side)
        with torch.cuda.stream(side):
            sdf_mid, radiance = query(lambda x, v: model.forward(x, v)[:2], pm, view)
    if cfg.calc_normal:
        sdf, nablas = query(model.forward_with_nablas, pts)
    else:
        sdf = query(model.forward_density_only, pts)[0]
    sdf = sdf.reshape(R, N)
    if side is not None:
        main.wait_stream(side)
        sdf_mid.record_stream(main)   # allocated on the side stream, consumed on the caller's from here on
        radiance.record_stream(main)
    else:
        sdf_mid, radiance = query(lambda x, v: model.forward(x, v)[:2], pm, view)
    d_final = dmid[:, :N - 1]
    if sdf.is_cuda and os.environ.get("NEUMESH_COMPOSITE", "hip") != "torch":
        rgb, depth, acc, normals, cdf, alpha, w = _HipComposite.apply(sdf, radiance, nablas if cfg.calc_normal else None, model.forward_s(), dmid,
                                                                       bool(cfg.white_bkgd))
        ret = OrderedDict(rgb=rgb, depth_volume=depth, mask_volume=acc)
        if cfg.calc_normal:
            ret["normals_volume"] = normals
    else:
        cdf, alpha = sdf_to_alpha(sdf, model.forward_s())
        w = alpha_to_w(alpha)
        rgb = torch.sum(w[..., None] * radiance, dim=-2)
        depth = torch.sum(w / (w.sum(-1, keepdim=True) + 1e-10) * d_final, dim=-1)
        acc = torch.sum(w, -1)
        if cfg.white_bkgd:
            rgb = rgb + (1.0 - acc[..., None])
        ret = OrderedDict(rgb=rgb, depth_volume=depth, mask_volume=acc)
        if cfg.calc_normal:
            nn_ = torch.nn.functional.normalize(nablas[:, :N - 1], dim=-1)
            ret["normals_volume"] = (nn_ * w[..., None]).sum(dim=-2)
    if detailed:
        if cfg.calc_normal:
            ret["implicit_nablas"] = nablas
        ret.update(implicit_surface=sdf, radiance=radiance, alpha=alpha, cdf=cdf, visibility_weights=w,
                   d_final=d_final, d_all=d, near_far=nf)
        if samples_output:
            ret.update(xyz=pm, dirs=dirn[:, None, :].expand(R, N - 1, 3), density=sdf_mid, colors=radiance)
    return ret


def volume_render(rays_o, rays_d, model, obj_bounding_radius=1.0, batched=False, batched_info={},
                  calc_normal=False, use_view_dirs=True, rayschunk=65536, netchunk=1048576, white_bkgd=False,
                  near_bypass: Optional[float] = None, far_bypass: Optional[float] = None, detailed_output=True,
                  show_progress=False, perturb=False, fixed_s_recp=1 / 64.0, N_samples=64, N_importance=64,
                  N_nograd_samples=2048, N_upsample_iters=4, samples_output=False, bounded_near_far=True,
                  random_color_direction=False, **dummy_kwargs):
    """Same contract as the reference's volume_render (models/renderer.py:105-368)."""
    if batched:
        B = rays_d.shape[0]
        lead = [B, -1]
    else:
        lead = [-1]
    if not use_view_dirs:
        raise NotImplementedError("neumesh_amd.volume_render: use_view_dirs=False (the NeuMesh colour branch always takes view "
                                  "directions: models/frameworks/neumesh/neumesh.py:239-260)")
    training = torch.is_grad_enabled()   # trainer.py:75-81: autograd through the field + compositing (perturb alone: fused, cfg.u_rand)
    # plain NeuMesh field, inference, the rays' own directions: one C call per chunk.  Per-sample outputs and random
    # colour directions (training-side options, trainer.py:70-79,139-146) go through the staged form.
    fused = (isinstance(model, NeuMesh) or fusable_edit_model(model)) and not training and not samples_output and not random_color_direction
    if fused and not (model if isinstance(model, NeuMesh) else model.main_model).fused_supported():
        fused = False   # a configuration the fused kernels refuse (W != 256, ...): staged renderer over the model's methods (inference_route)
    cfg = make_render_cfg(obj_bounding_radius, N_samples, N_importance, N_upsample_iters, bounded_near_far, calc_normal,
                          white_bkgd, near_bypass, far_bypass)
    progress = None
    if show_progress:
        try:
            from tqdm import tqdm
            progress = tqdm
        except ImportError:
            progress = None
    flat_o = torch.reshape(rays_o, [-1, 3]).float()
    flat_d = torch.reshape(rays_d, [-1, 3]).float()
    if fused:
        ret = render_rays_fused(model, flat_o, flat_d, cfg, rayschunk, detailed=detailed_output, progress=progress, perturb=perturb)
    else:   # wrapper model (editing tools): per-ray stages on HIP, field through the wrapper's methods
        ret = render_rays_staged(model, flat_o, flat_d, cfg, rayschunk, netchunk, detailed=detailed_output, progress=progress,
                                 differentiable=torch.is_grad_enabled(), perturb=perturb, samples_output=samples_output,
                                 random_color_direction=random_color_direction)
    for k in list(ret.keys()):
        v = ret[k]
        ret[k] = v.reshape(*lead, *v.shape[1:]) if batched else v
    return ret["rgb"], ret["depth_volume"], ret


class SingleRenderer(nn.Module):
    """models/renderer.py:371-377."""

    def __init__(self, model):
        super().__init__()
        self.model = model

    def forward(self, rays_o, rays_d, **kwargs):
        return volume_render(rays_o, rays_d, self.model, **kwargs)

    def synchronize(self) -> bool:
        """Not part of the reference's class: wait for every render issued so far (calls return while their kernels run) and
        report whether all of them stayed inside the fp16 range of the split-half kernels (False: a RuntimeWarning named the affected calls and
        the model now runs the fp32 kernels -- render those frames again)."""
        models = [self.model] if isinstance(self.model, NeuMesh) else [m for m in self.model.modules() if isinstance(m, NeuMesh)]
        ok = True
        for m in models:
            ok = m.synchronize_fp16_range() and ok
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        return ok
