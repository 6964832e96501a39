"""NeuMesh field -- host-side mirror of the reference's
``models/frameworks/neumesh/neumesh.py`` (class NeuMesh, :16-273) on the HIP library.

Same constructor arguments, parameter names / shapes (so the reference's checkpoints load with
``load_state_dict`` unchanged: SURVEY.md section 5) and the same public methods:

    forward(xyz, view_dirs, need_nablas=True, nablas_only=False, return_ds=False)
    forward_density_only(xyz) / forward_with_nablas(xyz) / forward_color(...) / forward_s()
    forward_indicator_weight() / compute_distance(xyz)

Under ``torch.no_grad()`` (how render.py calls the model, render.py:209) every method is one or
two fused HIP launches (``nm_field_density`` / ``nm_field_forward`` / ``nm_field_color``).  With
autograd enabled (training, editing fine-tunes) the K-NN still comes from the HIP kernel and
the remaining arithmetic is expressed with torch ops on the device, so first and second
derivatives behave as in the reference.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn
from torch import autograd

from . import _lib


class Embedder(nn.Module):
    """models/base.py:15-70 with get_embedder's settings (:73-87): [x, sin(x f), cos(x f), ...],
    f = 2**0 .. 2**(L-1)."""

    def __init__(self, input_dim: int, n_freqs: int):
        super().__init__()
        self.input_dim, self.n_freqs = input_dim, n_freqs
        self.out_dim = input_dim * (1 + 2 * n_freqs)

    def forward(self, x):
        out = [x]
        for j in range(self.n_freqs):
            f = float(2 ** j)
            out.append(torch.sin(x * f))
            out.append(torch.cos(x * f))
        return torch.cat(out, dim=-1)


def get_embedder(multires: int, input_dim: int = 3):
    if multires < 0:
        return nn.Identity(), input_dim
    e = Embedder(input_dim, multires)
    return e, e.out_dim


# mlp_precision -> nm_field_desc.mlp_precision.  "f16" = ONE f16 MFMA per product (plain fp16 operands, fp32 accumulation):
# the reduced-precision mode BASELINE configs[1] calls "bf16 MLP"; it misses the 1e-4 RGB bound and is never a default
# "f16x2+f16col": split-half geometry network (whose error the s = 400 sigmoid amplifies) + single-product colour network (not
# amplified: |d rgb| <= 1/4 |d z|); "f16x2s": the three products in ONE accumulator (unscaled residual halves, nm_mlp_h2.h);
# "f16x2s+f16col": both.  Their errors against the reference are in every bench line and gated in tests/test_gpu_parity.py.
_PRECISION_CODES = {"fp32": 0, "f16x2": 2, "f16": 4, "f16x2+f16col": 5, "f16x2s": 6, "f16x2s+f16col": 7}


def interpolation(features, indices, weights):
    """neumesh.py:11-13."""
    return torch.sum(features[indices] * weights.unsqueeze(-1), dim=-2)


def _mlp(in_dim: int, width: int, depth: int, act, wn: bool) -> nn.Sequential:
    """Sequential(Linear, act, Sequential(Linear, act), ...) -- the nesting that produces the
    reference's state-dict keys `<name>.0.*`, `<name>.2.0.*`, `<name>.3.0.*`, ..."""
    def lin(i, o):
        layer = nn.Linear(i, o)
        return nn.utils.weight_norm(layer) if wn else layer
    mods = [lin(in_dim, width), act()]
    for _ in range(depth - 1):
        mods.append(nn.Sequential(lin(width, width), act()))
    return nn.Sequential(*mods)


class _FusedField(autograd.Function):
    """Training-side form of the field queries: the FORWARD values come from the fused HIP
    kernels -- sdf, and nabla = d sdf / d xyz through the kernel's closed-form forward-mode tangent instead of
    an autograd.grad pass (neumesh.py:225-232 in the reference) -- and no graph of the ~60 torch ops per query
    is kept.  BACKWARD re-evaluates the same query with the torch-op restatement (`NeuMesh._forward_density` /
    `_forward_color`, K-NN again from the HIP kernel) under autograd and differentiates that: first derivatives
    of sdf / rgb and, for a cotangent on nabla (the eikonal loss), the second derivative of the nabla graph.
    Costs one extra forward per backward (as activation checkpointing does); saves the activation memory."""

    @staticmethod
    def forward(ctx, model, mode, xyz, view_dirs, *params):
        with torch.no_grad():
            if mode == "density":
                out = (model._fused_density(xyz, False)[0],)
            elif mode == "density_nabla":
                out = model._fused_density(xyz, True)
            else:   # "forward": sdf, rgb
                out = model._fused_forward(xyz, view_dirs, False)[:2]
        ctx.model, ctx.mode = model, mode
        ctx.save_for_backward(xyz, view_dirs if view_dirs is not None else xyz.new_zeros(0))
        ctx.n_params = len(params)
        return out if len(out) > 1 else out[0]

    @staticmethod
    def backward(ctx, *cotangents):
        model, mode = ctx.model, ctx.mode
        xyz, view_dirs = ctx.saved_tensors
        params = model._trainable()
        with torch.enable_grad():
            x = xyz.detach().requires_grad_(True)
            if mode == "density":
                outs = (model._density_autograd(x, False)[0],)
            elif mode == "density_nabla":
                outs = model._density_autograd(x, True)[:2]
            else:
                outs = model._forward_autograd(x, view_dirs.detach(), True, False, False)[:2]
            pairs = [(o, g) for o, g in zip(outs, cotangents) if g is not None and o.requires_grad]
            grads = autograd.grad([o for o, _ in pairs], [x] + params, [g for _, g in pairs], allow_unused=True) if pairs else \
                [None] * (1 + len(params))
        gx = grads[0] if ctx.needs_input_grad[2] else None
        return (None, None, gx, None, *grads[1:])


class _HipField(autograd.Function):
    """Training-side form of the field queries with BOTH directions on the HIP library (C ABI
    nm_train_forward / nm_train_backward, csrc/nm_train.h): forward keeps every intermediate in a device workspace, backward is the
    closed-form reverse pass -- nabla is the kernel's forward-mode tangent, so a cotangent on it (eikonal loss, normals, the
    colour MLP's nabla input) needs no create_graph=True pass (neumesh.py:223-232 in the reference) -- and no torch-op
    graph is built or replayed.  Tensor inputs (gradients returned for those that require them): weight-norm-FOLDED geometry
    weights (autograd maps their gradients on to g / v), biases, colour weights / biases, both code tables, the indicator
    vectors and the indicator weight w1 = sigmoid(raw)."""

    @staticmethod
    def forward(ctx, model, mode, xyz, view_dirs, *tensors):
        lib = _lib.load()
        q = xyz.detach().float().reshape(-1, 3).contiguous()
        dev = q.device
        tile = model._tile_order(xyz.shape, dev)
        v = None
        if mode == "forward":
            v = view_dirs.detach().float().expand_as(xyz).reshape(-1, 3).contiguous()
        if tile is not None:
            q = q[tile[0]]
            v = None if v is None else v[tile[0]]
        P = q.shape[0]
        desc, keep = model._train_desc(tensors)
        t = _lib.FieldTables()
        gf, cf, iv = (x.detach().float().contiguous() for x in tensors[-4:-1])
        t.geometry_features, t.color_features, t.indicator_vector = gf.data_ptr(), cf.data_ptr(), iv.data_ptr()
        t.indicator_weight, t.s = model._host_scalars()
        with_nabla = 0 if mode == "density" else 1
        f32 = dict(dtype=torch.float32, device=dev)
        sdf = torch.empty((P,), **f32)
        nab = torch.empty((P, 3), **f32) if with_nabla else None
        rgb = torch.empty((P, 3), **f32) if mode == "forward" else None
        ws = torch.empty((int(lib.nm_train_workspace_bytes(C.byref(desc), P)),), dtype=torch.uint8, device=dev)
        grid = model.grid_for(dev).grid.handle
        with torch.cuda.device(dev):
            _lib.check(lib.nm_train_forward(C.byref(desc), grid, C.byref(t), _lib.ptr(q), _lib.ptr(v), P, with_nabla, _lib.ptr(sdf), _lib.ptr(nab),
                                            _lib.ptr(rgb), _lib.ptr(ws), _lib.current_stream(dev)), "nm_train_forward")
        ctx.model, ctx.mode, ctx.ws, ctx.tile, ctx.P = model, mode, ws, tile, P
        ctx.desc, ctx.tables, ctx.keep = desc, t, (keep, gf, cf, iv, tensors)
        # backward re-reads the weights / tables through the pointers of `desc` and `t` (nothing is copied): remember the version counters
        # so that an in-place change in between (an optimizer step between two backward passes, weight clipping) is an error, as it is
        # for tensors autograd saves itself
        ctx.versions = [x._version for x in tensors]
        ctx.grid = grid
        ctx.shapes = [tuple(x.shape) for x in tensors]
        lead = xyz.shape[:-1]
        def back(a, w_):
            return (a if tile is None else a[tile[1]]).reshape(*lead, w_)
        if mode == "density":
            return back(sdf, 1)
        if mode == "density_nabla":
            return back(sdf, 1), back(nab, 3)
        return back(sdf, 1), back(rgb, 3)

    @staticmethod
    def backward(ctx, *cot):
        lib = _lib.load()
        model, mode, tile, P = ctx.model, ctx.mode, ctx.tile, ctx.P
        if ctx.ws is None:
            raise RuntimeError("_HipField: backward through the same field query a second time -- its workspace is released by the first "
                               "backward pass (retain_graph is not supported on the HIP training path; use autograd_backend='torch')")
        changed = [i for i, (x, v) in enumerate(zip(ctx.keep[4], ctx.versions)) if x._version != v]
        if changed:
            raise RuntimeError(f"_HipField: {len(changed)} of the field's parameter tensors were modified in place between forward and backward "
                               "(the backward pass reads them where they are); run backward before the optimizer step")
        dev = ctx.ws.device

        def fwd(g, w_):
            if g is None:
                return None
            g = g.detach().float().reshape(-1, w_)
            return (g if tile is None else g[tile[0]]).contiguous()
        g_sdf = fwd(cot[0], 1)
        g_nab = fwd(cot[1], 3) if mode == "density_nabla" else None
        g_rgb = fwd(cot[1], 3) if mode == "forward" else None
        n_geo, n_col = model._cfg["D_density"], model._cfg["D_color"]
        need = list(ctx.needs_input_grad[4:])
        if mode != "forward":     # geometry-only queries: the colour MLP and the colour table are not part of the graph
            for i in list(range(2 * (n_geo + 1), 2 * (n_geo + 1) + 2 * (n_col + 1))) + [len(need) - 3]:
                need[i] = False
        # one zeroed buffer for every wanted gradient (one fill launch instead of one per parameter); 64-float alignment per tensor
        sizes = [(-(-int(np.prod(shape)) // 64) * 64 if nd else 0) for shape, nd in zip(ctx.shapes, need)]
        flat = torch.zeros((sum(sizes),), dtype=torch.float32, device=dev)
        grads, off = [], 0
        for shape, nd, sz in zip(ctx.shapes, need, sizes):
            grads.append(flat[off:off + int(np.prod(shape))].view(shape) if nd else None)
            off += sz
        out = _lib.TrainGrads()
        it = iter(grads)
        def nxt():
            g = next(it)
            return None if g is None else g.data_ptr()
        for l in range(n_geo):
            out.geo_weight[l] = nxt()
        out.density_weight = nxt()
        for l in range(n_geo):
            out.geo_bias[l] = nxt()
        out.density_bias = nxt()
        for l in range(n_col):
            out.col_weight[l] = nxt()
        out.rgb_weight = nxt()
        for l in range(n_col):
            out.col_bias[l] = nxt()
        out.rgb_bias = nxt()
        out.geometry_features, out.color_features, out.indicator_vector, out.indicator_weight = nxt(), nxt(), nxt(), nxt()
        with torch.cuda.device(dev):
            _lib.check(lib.nm_train_backward(C.byref(ctx.desc), ctx.grid, C.byref(ctx.tables), P, 0 if mode == "density" else 1,
                                             1 if mode == "forward" else 0, _lib.ptr(g_sdf), _lib.ptr(g_nab), _lib.ptr(g_rgb), _lib.ptr(ctx.ws),
                                             C.byref(out), _lib.current_stream(dev)), "nm_train_backward")
        ctx.ws = None
        return (None, None, None, None, *grads)


class FieldHandle:
    """Owns one nm_field_t (packed MLP weights on one device).  NeuMesh keeps a reference; nn.DataParallel replicas are
    shallow copies of the module's __dict__ and therefore share THIS object until they rebuild their own (see
    NeuMesh._replicate_for_data_parallel): the handle is destroyed exactly once, when the last reference goes."""

    def __init__(self, h, device):
        self.h, self.device = h, device

    def __del__(self):
        try:
            if getattr(self, "h", None):
                _lib.load(require_device=False).nm_field_destroy(self.h)
                self.h = None
        except Exception:
            pass


class NeuMesh(nn.Module):
    def __init__(self, mesh_grid, D_density: int, D_color: int, W: int, geometry_dim: int, color_dim: int,
                 multires_view: int, multires_d: int, multires_fg: int, multires_ft: int,
                 enable_nablas_input: bool, input_view_dim=3, input_d_dim=1, ln_s=0.2996, speed_factor=1.0,
                 learn_indicator_weight=True):
        super().__init__()
        if input_view_dim != 3 or input_d_dim != 1:
            raise ValueError("NeuMesh: input_view_dim=3 and input_d_dim=1 are the only supported values")
        self.mesh_grid = mesh_grid
        num_vertices = mesh_grid.get_number_of_vertices()
        self.ln_s = nn.Parameter(torch.tensor([float(ln_s)]))
        self.speed_factor = speed_factor
        self.geometry_features = nn.Parameter(torch.randn(num_vertices, geometry_dim))
        self.color_features = nn.Parameter(torch.randn(num_vertices, color_dim))
        self.indicator_vector = nn.Parameter(mesh_grid.get_vertex_normal_torch().float().clone())
        self.learn_indicator_weight = learn_indicator_weight
        if learn_indicator_weight:
            self.indicator_weight_raw = nn.Parameter(torch.tensor([-2.0]))
        self.embed_fn_d, ch_d = get_embedder(multires_d, 1)
        self.embed_fn_view, ch_view = get_embedder(multires_view, 3)
        self.embed_fn_fg, ch_fg = get_embedder(multires_fg, geometry_dim)
        self.embed_fn_ft, ch_ft = get_embedder(multires_ft, color_dim)
        self.softplus = nn.Softplus(beta=100)
        self.pts_linears = _mlp(ch_d + ch_fg, W, D_density, lambda: self.softplus, wn=True)
        self.enable_nablas_input = enable_nablas_input
        ch_color = ch_view + ch_ft + ch_d + (3 if enable_nablas_input else 0)
        self.views_linears = _mlp(ch_color, W, D_color, lambda: nn.ReLU(inplace=True), wn=False)
        self.density_linear = nn.utils.weight_norm(nn.Linear(W, 1))
        self.color_linear = nn.Sequential(nn.Linear(W, 3), nn.Sigmoid())
        self._cfg = dict(W=W, D_density=D_density, D_color=D_color, geometry_dim=geometry_dim, color_dim=color_dim,
                         multires_d=multires_d, multires_fg=multires_fg, multires_ft=multires_ft,
                         multires_view=multires_view)
        # MLP arithmetic of the fused HIP path: "f16x2s" (default: split-half f16 MFMA -- every operand carried as an fp16 value plus
        # an fp16 residual, three products into one fp32 accumulator; as accurate as the fp32 form against the reference, 2.2-2.5x
        # faster; needs |activations| < 65504), "f16x2" (the same with the residual halves scaled by 2^11 and a second accumulator: the
        # default of rounds 1-3, 3-7 % slower per kernel), "fp32" (fp32-input MFMA), "f16" (single f16 product: reduced precision,
        # error-quantified in bench.py) or "...+f16col" (single product in the colour network only).
        self.mlp_precision = os.environ.get("NEUMESH_MLP_PRECISION", "f16x2s")
        self._field = None        # FieldHandle (owner of the nm_field_t)
        self._field_key = None    # parameter versions / device / precision the packed weights were built from
        self._field_dev = None
        self._field_epoch = 0     # bumped by invalidate_field()
        self._range_checked = False
        self._range_pending = []  # deferred fp16-range reads: (event, pinned host word) per fused call that returned without a sync
        self._range_words = None  # 64 pinned int32 words handed out in rotation (allocated at the first deferred read)
        # With autograd enabled (training): "hip" = forward AND backward on the HIP library (_HipField: nm_train_forward /
        # nm_train_backward, closed-form reverse pass); "recompute" = fused HIP forward + a backward that re-evaluates the
        # torch-op restatement (_FusedField); "torch" = the torch-op restatement end to end.  Same gradients.
        self.autograd_backend = os.environ.get("NEUMESH_AUTOGRAD", "hip")
        self._keep = None         # tensors whose pointers the last FieldDesc referenced

    # ------------------------------------------------------------------ scalars
    def forward_s(self):
        return torch.exp(self.ln_s * self.speed_factor)

    def forward_indicator_weight(self):
        return torch.sigmoid(self.indicator_weight_raw)

    def _w1(self):
        return self.forward_indicator_weight() if self.learn_indicator_weight else 0.1

    # ------------------------------------------------------------------ HIP field handle
    def _geo_layers(self):
        return [self.pts_linears[0]] + [self.pts_linears[i][0] for i in range(2, len(self.pts_linears))]

    def _col_layers(self):
        return [self.views_linears[0]] + [self.views_linears[i][0] for i in range(2, len(self.views_linears))]

    def _mlp_params(self):
        ps = []
        for m in self._geo_layers() + [self.density_linear]:
            ps += [m.weight_g, m.weight_v, m.bias]
        for m in self._col_layers() + [self.color_linear[0]]:
            ps += [m.weight, m.bias]
        return ps

    def field_handle(self):
        """nm_field_t with the current MLP weights (weight-norm folded), re-packed when they change."""
        ps = self._mlp_params()
        if self.mlp_precision not in _PRECISION_CODES:
            raise ValueError(f"mlp_precision={self.mlp_precision!r}: expected one of {sorted(_PRECISION_CODES)}")
        dev = ps[0].device
        key = tuple((p.data_ptr(), p._version) for p in ps) + (self.mlp_precision, str(dev), self._field_epoch)
        if self._field is not None and key == self._field_key:
            return self._field.h
        lib = _lib.load()
        if dev.type != "cuda":
            raise _lib.NeuMeshHipError("NeuMesh parameters must live on a HIP device (model.to('cuda')); no CPU fallback")
        if self._field is not None and self._field.device != dev:
            # the packed weights live on the device they were created on: a moved model gets a new handle (the old one
            # goes with its last owner)
            self._field = None
        with torch.no_grad():
            def folded(m):  # W = g * v / ||v||_row  (torch.nn.utils.weight_norm, dim=0)
                return (m.weight_v * (m.weight_g / m.weight_v.norm(dim=1, keepdim=True))).float().contiguous()
            gw = [folded(m) for m in self._geo_layers()]
            gb = [m.bias.detach().float().contiguous() for m in self._geo_layers()]
            dw, db = folded(self.density_linear), self.density_linear.bias.detach().float().contiguous()
            cw = [m.weight.detach().float().contiguous() for m in self._col_layers()]
            cb = [m.bias.detach().float().contiguous() for m in self._col_layers()]
            rw = self.color_linear[0].weight.detach().float().contiguous()
            rb = self.color_linear[0].bias.detach().float().contiguous()
        d = _lib.FieldDesc()
        c = self._cfg
        d.W, d.D_density, d.D_color = c["W"], c["D_density"], c["D_color"]
        d.geometry_dim, d.color_dim = c["geometry_dim"], c["color_dim"]
        d.multires_d, d.multires_fg, d.multires_ft, d.multires_view = c["multires_d"], c["multires_fg"], c["multires_ft"], c["multires_view"]
        d.enable_nablas_input, d.use_view_dirs = int(self.enable_nablas_input), 1
        d.mlp_precision = _PRECISION_CODES[self.mlp_precision]
        for i, (w_, b_) in enumerate(zip(gw, gb)):
            d.geo_weight[i], d.geo_bias[i] = w_.data_ptr(), b_.data_ptr()
        for i, (w_, b_) in enumerate(zip(cw, cb)):
            d.col_weight[i], d.col_bias[i] = w_.data_ptr(), b_.data_ptr()
        d.density_weight, d.density_bias = dw.data_ptr(), db.data_ptr()
        d.rgb_weight, d.rgb_bias = rw.data_ptr(), rb.data_ptr()
        with torch.cuda.device(dev):
            stream = _lib.current_stream(dev)
            if self._field is None:
                h = C.c_void_p()
                _lib.check(lib.nm_field_create(C.byref(d), stream, C.byref(h)), "nm_field_create")
                self._field = FieldHandle(h, dev)
            else:
                _lib.check(lib.nm_field_update(self._field.h, C.byref(d), stream), "nm_field_update")
        self._keep = (gw, gb, dw, db, cw, cb, rw, rb)  # the pack call synchronised; kept for clarity
        self._field_key = key
        self._field_dev = dev
        self._range_checked = False   # new weights: the next fused call verifies the fp16 range once
        return self._field.h

    def _replicate_for_data_parallel(self):
        """nn.DataParallel replica (models/trainer.py:39-42 wraps the renderer when several device_ids are given): a shallow
        copy that must not share device state with the module it was copied from -- its packed weights are rebuilt on the
        replica's own device at first use, and its mesh index is the per-device copy MeshGrid.on_device() keeps."""
        replica = super()._replicate_for_data_parallel()
        replica._field, replica._field_key, replica._field_dev = None, None, None
        replica._scalars_key, replica._keep, replica._range_checked = None, None, False
        replica._range_pending, replica._range_words = [], None      # (its own pinned words: replicas run on their own threads)
        replica._is_replica = True
        return replica

    def grid_for(self, device):
        """The mesh index on `device` (the module's own grid, or its per-device copy for a DataParallel replica)."""
        g = self.mesh_grid
        return g if g.device == device else g.on_device(device)

    def check_fp16_range(self, force: bool = False, every: int = 1, deferred_calls: int = 0) -> bool:
        """Split-half modes only: ask the library whether any launch since the last check saw a value
        outside the fp16 range (nm_field_overflow; synchronises the stream).  Called once after the
        first fused call on a new weight set; returns True if the results of those launches are valid.
        On overflow the model switches itself to mlp_precision='fp32' (with a warning) and returns False:
        the caller re-runs the call."""
        # The flag is sticky on the device and depends on the INPUTS as well as the weights (hidden activations at other views,
        # nabla magnitudes), so it is read after EVERY fused render call (every = 1: one stream sync per frame-sized call) and,
        # for the point-wise field calls, after the first call on a weight set and then every `every`-th call (a training step
        # issues a dozen of them; an overflow is then reported at most `every` calls late -- the flag does not forget).
        if self.mlp_precision == "fp32" or self._field is None:
            return True
        self._range_calls = getattr(self, "_range_calls", 0) + 1
        if self._range_checked and not force and every > 1 and self._range_calls % every:
            return True
        lib = _lib.load()
        flag = C.c_int(0)
        with torch.cuda.device(self._field_dev):
            _lib.check(lib.nm_field_overflow(self._field.h, C.byref(flag), _lib.current_stream(self._field_dev)), "nm_field_overflow")
        self._range_checked = True
        if not flag.value:
            self._range_last_read = self._range_calls
            return True
        import warnings
        late = self._range_calls - getattr(self, "_range_last_read", 0)
        warnings.warn("NeuMesh: an MLP activation or input left the fp16 range (|v| >= 65504) in the split-half f16 mode; "
                      + ("switching this model to mlp_precision='fp32' and re-running this call" if not deferred_calls else
                         f"found by the deferred (sync-free) check: the outputs of up to {deferred_calls} EARLIER render call(s) on this model may hold "
                         "Inf / NaN-affected values -- render them again; this model now runs mlp_precision='fp32' (NEUMESH_EAGER_RANGE_CHECK=1 checks "
                         "every call before it returns)")
                      + (f" -- the flag is polled every {every} point-wise calls, so up to {late - 1} EARLIER calls since the last poll may hold "
                         "Inf / NaN-affected values: repeat the enclosing render / ray-casting call" if late > 1 else ""), RuntimeWarning)
        self.mlp_precision = "fp32"
        return False

    def post_fp16_range_check(self) -> None:
        """The sync-free form of check_fp16_range (nm_field_overflow_post, ABI v11): queue a copy of the device flag into a pinned host word
        on the current stream, record an event behind it and return.  poll_fp16_range() reads the words whose events have completed -- the
        renderer does so at its next entry, `synchronize_fp16_range()` on demand."""
        if self.mlp_precision == "fp32" or self._field is None:
            return
        lib = _lib.load()
        if self._range_words is None:
            self._range_words, self._range_slot = torch.zeros(64, dtype=torch.int32).pin_memory(), 0   # one pinned allocation per model, 64 words in rotation
        if len(self._range_pending) >= 48:    # a caller that never comes back to poll: read what is ready, wait for the oldest if none is
            self.poll_fp16_range(wait_oldest=len(self._range_pending) >= 63)
        word = self._range_words[self._range_slot:self._range_slot + 1]
        self._range_slot = (self._range_slot + 1) % 64
        word.zero_()
        with torch.cuda.device(self._field_dev):
            _lib.check(lib.nm_field_overflow_post(self._field.h, C.c_void_p(word.data_ptr()), _lib.current_stream(self._field_dev)), "nm_field_overflow_post")
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._field_dev))
        self._range_pending.append((ev, word))

    def poll_fp16_range(self, wait: bool = False, wait_oldest: bool = False) -> bool:
        """Read the deferred fp16-range words that are ready (all of them with wait=True: blocks until the calls that posted them are done).
        Returns False -- after a warning, with the model switched to mlp_precision='fp32' -- if one of those EARLIER calls saw a value outside
        the fp16 range: its outputs may hold Inf / NaN-affected values and must be rendered again."""
        bad, keep = False, []
        for i, (ev, word) in enumerate(self._range_pending):
            if wait or (wait_oldest and i == 0):
                ev.synchronize()
            if ev.query():
                bad = bad or bool(int(word[0]))
            else:
                keep.append((ev, word))
        n_read = len(self._range_pending) - len(keep)
        self._range_pending = keep
        if not bad:
            return True
        self._range_pending = []
        if self.mlp_precision != "fp32":
            self.check_fp16_range(force=True, deferred_calls=n_read)      # (synchronises, resets the device flag, warns, switches to fp32)
        return False

    def synchronize_fp16_range(self) -> bool:
        """Wait for every fused call issued so far on this model and report whether all of them stayed inside the fp16 range."""
        return self.poll_fp16_range(wait=True)

    def invalidate_field(self):
        """Force a re-pack of the MLP weights at the next use.  Needed only after edits that bypass
        autograd's version counters (``p.data.copy_()`` / ``p.data.mul_()``); ordinary in-place ops,
        optimizer steps and load_state_dict are detected automatically."""
        self._field_epoch += 1

    def field_tables(self, color_features=None):
        """nm_field_tables for the current codes / scalars (borrowed pointers; returns the struct
        and the tensors that must stay alive while it is in use)."""
        gf = self.geometry_features.detach().float().contiguous()
        cf = (self.color_features if color_features is None else color_features).detach().float().contiguous()
        iv = self.indicator_vector.detach().float().contiguous()
        t = _lib.FieldTables()
        t.geometry_features, t.color_features, t.indicator_vector = gf.data_ptr(), cf.data_ptr(), iv.data_ptr()
        t.indicator_weight, t.s = self._host_scalars()
        return t, (gf, cf, iv)

    def _host_scalars(self):
        """(indicator weight, s) as Python floats.  Reading a device scalar synchronises the stream, and every fused field
        call needs the pair: cached on the parameters' version counters (an optimizer step or load_state_dict bumps them;
        invalidate_field() covers edits through .data), so a training step pays for it once instead of once per call."""
        ps = [self.ln_s] + ([self.indicator_weight_raw] if self.learn_indicator_weight else [])
        key = tuple((p.data_ptr(), p._version) for p in ps) + (self._field_epoch,)
        if getattr(self, "_scalars_key", None) != key:
            with torch.no_grad():
                if self.learn_indicator_weight:
                    both = torch.stack([self.forward_indicator_weight().reshape(()), self.forward_s().reshape(())]).tolist()   # one transfer
                    self._scalars = (float(both[0]), float(both[1]))
                else:
                    self._scalars = (0.1, float(self.forward_s()))
            self._scalars_key = key
        return self._scalars

    # ------------------------------------------------------------------ which kernels can serve this configuration
    def fused_supported(self) -> bool:
        """The fused inference kernels (nm_field_* / nm_render_rays: LDS tiles of 256 columns, csrc/nm_api.hip:nm_field_validate) are
        built for hidden width 256 -- the reference's value (models/frameworks/neumesh/__init__.py:26) -- code widths <= 64 and
        MLP inputs <= 256 columns."""
        c = self._cfg
        in_geo = 1 + 2 * c["multires_d"] + c["geometry_dim"] * (1 + 2 * c["multires_fg"])
        in_col = (3 if self.enable_nablas_input else 0) + 1 + 2 * c["multires_d"] + 3 * (1 + 2 * c["multires_view"]) + c["color_dim"] * (1 + 2 * c["multires_ft"])
        return (c["W"] == 256 and all(4 <= c[k] <= 64 and c[k] % 4 == 0 for k in ("geometry_dim", "color_dim"))
                and all(c[k] >= 0 for k in ("multires_d", "multires_fg", "multires_ft", "multires_view")) and c["multires_d"] <= 16
                and c["multires_view"] <= 16 and in_geo <= 256 and in_col <= 256 and 1 <= c["D_density"] <= 8 and 1 <= c["D_color"] <= 8)

    def train_kernels_supported(self) -> bool:
        """The any-width kernels of the training path (nm_train_forward / nm_train_backward, csrc/nm_api.hip:nm_train_validate):
        hidden width a multiple of 16, code widths multiples of 4, <= 8 layers, 0..15 embedder bands."""
        c = self._cfg
        return (c["W"] >= 16 and c["W"] % 16 == 0 and 1 <= c["D_density"] <= 8 and 1 <= c["D_color"] <= 8
                and all(c[k] >= 4 and c[k] % 4 == 0 for k in ("geometry_dim", "color_dim"))
                and all(0 <= c[k] <= 15 for k in ("multires_d", "multires_fg", "multires_ft", "multires_view")))

    def inference_route(self) -> str:
        """'fused' (the inference kernels), 'general' (a configuration they refuse -- any other hidden width, wider codes: the
        reference takes W as a free constructor argument, neumesh.py:16-36 -- served by nm_train_forward, forward only, its workspace
        dropped after the call; the renderer then takes its staged form), or 'torch' (neither kernel set: device torch ops around the
        HIP K-NN).  One warning per model when the route is not 'fused'."""
        route = "fused" if self.fused_supported() else ("general" if self.train_kernels_supported() else "torch")
        if route != "fused" and not getattr(self, "_route_warned", False):
            import warnings
            c = self._cfg
            warnings.warn(f"NeuMesh(W={c['W']}, geometry_dim={c['geometry_dim']}, color_dim={c['color_dim']}, ...): outside the fused inference "
                          f"kernels' configuration (W = 256, codes <= 64, inputs <= 256 columns); inference runs on the "
                          + ("any-width kernels of the training path (nm_train_forward) and the staged renderer" if route == "general"
                             else "torch-op form of the field"), RuntimeWarning)
            self._route_warned = True
        return route

    def _general_query(self, mode, xyz, view_dirs=None, want_ds=False):
        """forward-only nm_train_forward for configurations the fused kernels refuse.  mode: 'density' | 'density_nabla' | 'forward'."""
        lib = _lib.load()
        q = xyz.detach().float().reshape(-1, 3).contiguous()
        dev = q.device
        tile = self._tile_order(xyz.shape, dev)
        v = view_dirs.detach().float().expand_as(xyz).reshape(-1, 3).contiguous() if mode == "forward" else None
        if tile is not None:
            q = q[tile[0]]
            v = None if v is None else v[tile[0]]
        P = q.shape[0]
        with torch.no_grad():
            tensors = self._train_tensors()
        desc, keep = self._train_desc(tensors)
        t, keep_t = self.field_tables()
        f32 = dict(dtype=torch.float32, device=dev)
        with_nabla = 0 if mode == "density" else 1
        sdf = torch.empty((P,), **f32)
        nab = torch.empty((P, 3), **f32) if with_nabla else None
        rgb = torch.empty((P, 3), **f32) if mode == "forward" else None
        ws = torch.empty((int(lib.nm_train_workspace_bytes(C.byref(desc), P)),), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.nm_train_forward(C.byref(desc), self.grid_for(dev).grid.handle, C.byref(t), _lib.ptr(q), _lib.ptr(v), P, with_nabla,
                                            _lib.ptr(sdf), _lib.ptr(nab), _lib.ptr(rgb), _lib.ptr(ws), _lib.current_stream(dev)), "nm_train_forward")
        del ws, keep, keep_t
        lead = xyz.shape[:-1]

        def back(a, w_):
            return None if a is None else (a if tile is None else a[tile[1]]).reshape(*lead, w_)
        out = (back(sdf, 1), back(rgb, 3), back(nab, 3))
        if want_ds:   # (the K-NN lists as the reference returns them: one more search, on the plain distance kernel)
            out = out + self.compute_distance(xyz.detach())
        return out

    # ------------------------------------------------------------------ fused (no-grad) paths
    _tile_cache = {}

    @classmethod
    def _tile_order(cls, shape, device):
        """Points that arrive as [..., rays, samples, 3] (the reference renderer's and ray caster's layout) are handed to the
        kernels as tiles of 16 adjacent rays x 4 consecutive samples: 64 consecutive points are then a compact packet for the
        wave-cooperative K-NN search, where ray-major order strings a wave's 64 queries along one ray.  Returns (perm, inv)
        index tensors over the flattened points, or None (fewer than 16 rays / 4 samples, flat point lists).  Every point's
        result is independent of the order, so this only changes speed (neumesh_amd.renderer does the same for its stages)."""
        if len(shape) < 3 or os.environ.get("NEUMESH_NO_TILE_ORDER"):
            return None
        P = int(shape[-2])
        R = 1
        for d in shape[:-2]:
            R *= int(d)
        if R < 16 or P < 4:
            return None
        key = (R, P, str(device))
        hit = cls._tile_cache.get(key)
        if hit is not None:
            return hit
        idx = torch.arange(R * P, device=device).view(R, P)
        R16, P4 = R // 16 * 16, P // 4 * 4
        perm = torch.cat([idx[:R16, :P4].reshape(R16 // 16, 16, P4 // 4, 4).permute(0, 2, 1, 3).reshape(-1), idx[:R16, P4:].reshape(-1), idx[R16:, :].reshape(-1)])
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(R * P, device=device)
        if R * P <= (1 << 22):   # (index tensors of big calls are rebuilt each time: 16 bytes per point are not worth keeping)
            if len(cls._tile_cache) > 16:
                cls._tile_cache.clear()
            cls._tile_cache[key] = (perm, inv)
        return perm, inv

    def _fused_density(self, xyz, want_nabla: bool):
        lib = _lib.load()
        q = xyz.detach().float().reshape(-1, 3).contiguous()
        tile = self._tile_order(xyz.shape, q.device)
        if tile is not None:
            q = q[tile[0]]
        P = q.shape[0]
        sdf = torch.empty((P,), dtype=torch.float32, device=q.device)
        nab = torch.empty((P, 3), dtype=torch.float32, device=q.device) if want_nabla else None
        scratch = torch.empty((int(lib.nm_field_scratch_bytes(P)),), dtype=torch.uint8, device=q.device)
        t, keep = self.field_tables()
        with torch.cuda.device(q.device):
            for _attempt in range(2):
                _lib.check(lib.nm_field_density(self.field_handle(), self.grid_for(q.device).grid.handle, C.byref(t), _lib.ptr(q), P,
                                                _lib.ptr(sdf), _lib.ptr(nab), _lib.ptr(scratch), _lib.current_stream(q.device)),
                           "nm_field_density")
                if self.check_fp16_range(every=64):
                    break
        del keep
        if tile is not None:
            sdf, nab = sdf[tile[1]], (None if nab is None else nab[tile[1]])
        return sdf.reshape(*xyz.shape[:-1], 1), (None if nab is None else nab.reshape(xyz.shape))

    def _fused_forward(self, xyz, view_dirs, want_ds: bool):
        lib = _lib.load()
        q = xyz.detach().float().reshape(-1, 3).contiguous()
        v = view_dirs.detach().float().expand_as(xyz).reshape(-1, 3).contiguous()
        tile = self._tile_order(xyz.shape, q.device)
        if tile is not None:
            q, v = q[tile[0]], v[tile[0]]
        P, dev = q.shape[0], q.device
        sdf = torch.empty((P,), dtype=torch.float32, device=dev)
        rgb = torch.empty((P, 3), dtype=torch.float32, device=dev)
        nab = torch.empty((P, 3), dtype=torch.float32, device=dev)
        ds = torch.empty((P,), dtype=torch.float32, device=dev) if want_ds else None
        idx = torch.empty((P, 8), dtype=torch.int64, device=dev) if want_ds else None
        w = torch.empty((P, 8), dtype=torch.float32, device=dev) if want_ds else None
        scratch = torch.empty((int(lib.nm_field_scratch_bytes(P)),), dtype=torch.uint8, device=dev)
        t, keep = self.field_tables()
        with torch.cuda.device(dev):
            for _attempt in range(2):
                _lib.check(lib.nm_field_forward(self.field_handle(), self.grid_for(q.device).grid.handle, C.byref(t), _lib.ptr(q), _lib.ptr(v), P,
                                                _lib.ptr(sdf), _lib.ptr(rgb), _lib.ptr(nab), _lib.ptr(ds), _lib.ptr(idx), _lib.ptr(w),
                                                _lib.ptr(scratch), _lib.current_stream(dev)), "nm_field_forward")
                if self.check_fp16_range(every=64):
                    break
        del keep
        if tile is not None:
            sdf, rgb, nab = sdf[tile[1]], rgb[tile[1]], nab[tile[1]]
            if want_ds:
                ds, idx, w = ds[tile[1]], idx[tile[1]], w[tile[1]]
        lead = xyz.shape[:-1]
        out = (sdf.reshape(*lead, 1), rgb.reshape(*lead, 3), nab.reshape(*lead, 3))
        if want_ds:
            out = out + (ds.reshape(*lead, 1), idx.reshape(*lead, 8), w.reshape(*lead, 8))
        return out

    # ------------------------------------------------------------------ HIP training path (_HipField)
    def _train_tensors(self):
        """Tensor inputs of _HipField, in its fixed order: folded geometry weights (+ density head), their biases, colour weights
        (+ rgb head), their biases, geometry / colour tables, indicator vectors, w1.  The folding W = v * (g / |v|_row)
        (torch.nn.utils.weight_norm, dim 0) is done here with torch ops so that autograd carries the gradient on to g and v."""
        def folded(m):
            return m.weight_v * (m.weight_g / m.weight_v.norm(dim=1, keepdim=True))
        geo, col = self._geo_layers() + [self.density_linear], self._col_layers() + [self.color_linear[0]]
        w1 = self.forward_indicator_weight() if self.learn_indicator_weight else self.ln_s.new_tensor([0.1])
        return ([folded(m) for m in geo] + [m.bias for m in geo] + [m.weight for m in col] + [m.bias for m in col] +
                [self.geometry_features, self.color_features, self.indicator_vector, w1])

    def _train_desc(self, tensors):
        """nm_field_desc over the fp32 weights in `tensors` (the order of _train_tensors); returns (desc, tensors to keep alive)."""
        c = self._cfg
        ng, nc = c["D_density"], c["D_color"]
        flat = [x.detach().float().contiguous() for x in tensors[:2 * (ng + 1) + 2 * (nc + 1)]]
        gw, gb = flat[:ng + 1], flat[ng + 1:2 * (ng + 1)]
        cw, cb = flat[2 * (ng + 1):2 * (ng + 1) + nc + 1], flat[2 * (ng + 1) + nc + 1:]
        d = _lib.FieldDesc()
        d.W, d.D_density, d.D_color = c["W"], ng, nc
        d.geometry_dim, d.color_dim = c["geometry_dim"], c["color_dim"]
        d.multires_d, d.multires_fg, d.multires_ft, d.multires_view = c["multires_d"], c["multires_fg"], c["multires_ft"], c["multires_view"]
        # mlp_precision of a training descriptor selects the GEMMs of nm_train_*: 0 = the fp32 matrix pipe, anything else = the bf16 x 3 form
        # (fp32-grade operands cut into three bf16 pieces, csrc/nm_gemm.h); NEUMESH_TRAIN_GEMM=fp32 keeps the fp32 pipe for A/B
        fp32_gemm = self.mlp_precision == "fp32" or os.environ.get("NEUMESH_TRAIN_GEMM", "bf16x3") == "fp32"
        d.enable_nablas_input, d.use_view_dirs, d.mlp_precision = int(self.enable_nablas_input), 1, (0 if fp32_gemm else 6)
        for l in range(ng):
            d.geo_weight[l], d.geo_bias[l] = gw[l].data_ptr(), gb[l].data_ptr()
        d.density_weight, d.density_bias = gw[ng].data_ptr(), gb[ng].data_ptr()
        for l in range(nc):
            d.col_weight[l], d.col_bias[l] = cw[l].data_ptr(), cb[l].data_ptr()
        d.rgb_weight, d.rgb_bias = cw[nc].data_ptr(), cb[nc].data_ptr()
        return d, flat

    # ------------------------------------------------------------------ public API (reference names)
    def compute_distance(self, xyz):
        """neumesh.py:262-273."""
        ds, indices, weights = self.grid_for(xyz.device).compute_distance(
            xyz.reshape(-1, 3), indicator_vector=self.indicator_vector, indicator_weight=self._w1())
        lead = xyz.shape[:-1]
        return ds.reshape(*lead, -1), indices.reshape(*lead, -1), weights.reshape(*lead, -1)

    def forward_density_only(self, xyz):
        """neumesh.py:140-145."""
        if not torch.is_grad_enabled():
            route = self.inference_route()
            if route == "fused":
                return self._fused_density(xyz, False)[0]
            if route == "general":
                return self._general_query("density", xyz)[0]
            return self._density_autograd(xyz, False)[0]
        if self._autograd_backend() == "hip" and not xyz.requires_grad:
            return _HipField.apply(self, "density", xyz, None, *self._train_tensors())
        if self._autograd_backend() == "recompute":
            return _FusedField.apply(self, "density", xyz, None, *self._trainable())
        return self._density_autograd(xyz, False)[0]

    def forward_with_nablas(self, xyz):
        """neumesh.py:147-154."""
        if not torch.is_grad_enabled():
            route = self.inference_route()
            if route == "fused":
                return self._fused_density(xyz, True)
            if route == "general":
                sdf, _, nab = self._general_query("density_nabla", xyz)
                return sdf, nab
            return self._density_autograd(xyz, True)[:2]
        if self._autograd_backend() == "hip" and not xyz.requires_grad:
            return _HipField.apply(self, "density_nabla", xyz, None, *self._train_tensors())
        if self._autograd_backend() == "recompute":
            return _FusedField.apply(self, "density_nabla", xyz, None, *self._trainable())
        return self._density_autograd(xyz, True)[:2]

    def forward(self, xyz, view_dirs, need_nablas=True, nablas_only=False, return_ds=False):
        """neumesh.py:113-138."""
        if not torch.is_grad_enabled() and need_nablas and self.inference_route() != "torch":
            query = self._fused_forward if self.inference_route() == "fused" else (lambda x, v, ds_: self._general_query("forward", x, v, ds_))
            sdf, rgb, nab, *rest = query(xyz, view_dirs, return_ds)
            out = (sdf, nab) if nablas_only else (sdf, rgb)
            return out + tuple(rest)
        if torch.is_grad_enabled() and need_nablas and not nablas_only and not return_ds:
            if self._autograd_backend() == "hip" and not xyz.requires_grad and not view_dirs.requires_grad:
                return _HipField.apply(self, "forward", xyz, view_dirs, *self._train_tensors())
            if self._autograd_backend() == "recompute":
                return _FusedField.apply(self, "forward", xyz, view_dirs.expand_as(xyz), *self._trainable())
        return self._forward_autograd(xyz, view_dirs, need_nablas, nablas_only, return_ds)

    def _autograd_backend(self) -> str:
        """`autograd_backend`, except that a configuration nm_train_validate refuses (hidden width not a multiple of 16, code widths
        not multiples of 4, > 8 layers, > 15 bands) trains on the torch-op form instead of raising inside a query."""
        if self.autograd_backend == "hip" and not self.train_kernels_supported():
            if not getattr(self, "_backend_warned", False):
                import warnings
                warnings.warn("NeuMesh: this configuration is outside the HIP training kernels (nm_train_validate); autograd runs on the "
                              "torch-op form of the field (autograd_backend='torch')", RuntimeWarning)
                self._backend_warned = True
            return "torch"
        if self.autograd_backend == "recompute" and not self.fused_supported():
            return "torch"
        return self.autograd_backend

    # ------------------------------------------------------------------ torch-op (autograd) forms of the queries
    def _trainable(self):
        return [p for p in self.parameters() if p.requires_grad]

    def _density_autograd(self, xyz, need_nablas):
        """forward_density_only / forward_with_nablas with autograd (neumesh.py:140-154)."""
        if need_nablas:
            xyz.requires_grad_(True)
        with (torch.enable_grad() if need_nablas else contextlib.nullcontext()):
            ds, indices, weights = self.compute_distance(xyz)
        return self._forward_density(xyz, ds, self.geometry_features, indices, weights, need_nablas=need_nablas)

    def _forward_autograd(self, xyz, view_dirs, need_nablas, nablas_only, return_ds):
        """NeuMesh.forward with autograd (neumesh.py:113-138, 176-202)."""
        if need_nablas:
            xyz.requires_grad_(True)
        with (torch.enable_grad() if need_nablas else contextlib.nullcontext()):
            ds, indices, weights = self.compute_distance(xyz)
        density, nablas, d_emb = self._forward_density(xyz, ds, self.geometry_features, indices, weights, need_nablas=need_nablas)
        out = (density, nablas) if nablas_only else (
            density, self._forward_color(d_emb, view_dirs, self.color_features, indices, weights, nablas))
        if return_ds:
            out = out + (ds, indices, weights)
        return out

    def forward_color(self, d, view_dirs, color_features, indices=None, weights=None, nabla=None):
        """neumesh.py:156-168."""
        if not torch.is_grad_enabled() and self.fused_supported():
            lib = _lib.load()
            lead = d.shape[:-1]
            dd = d.detach().float().reshape(-1).contiguous()
            P, dev = dd.shape[0], dd.device
            v = view_dirs.detach().float().reshape(-1, 3).contiguous()
            ii = indices.detach().reshape(-1, 8).to(torch.int64).contiguous()
            ww = weights.detach().float().reshape(-1, 8).contiguous()
            nn_ = None if nabla is None else nabla.detach().float().reshape(-1, 3).contiguous()
            cf = color_features.detach().float().contiguous()
            rgb = torch.empty((P, 3), dtype=torch.float32, device=dev)
            scratch = torch.empty((int(lib.nm_field_scratch_bytes(P)),), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                for _attempt in range(2):
                    _lib.check(lib.nm_field_color(self.field_handle(), _lib.ptr(cf), _lib.ptr(dd), _lib.ptr(v), _lib.ptr(ii), _lib.ptr(ww),
                                                  _lib.ptr(nn_), P, _lib.ptr(rgb), _lib.ptr(scratch), _lib.current_stream(dev)),
                               "nm_field_color")
                    if self.check_fp16_range(every=64):
                        break
            return rgb.reshape(*lead, 3)
        return self._forward_color(self.embed_fn_d(d), view_dirs, color_features, indices, weights, nabla)

    # ------------------------------------------------------------------ autograd (torch-op) paths
    def _forward_density(self, xyz, d, geometry_features, indices, weights, need_nablas=False):
        """neumesh.py:204-237 in torch ops (used when autograd is on)."""
        with (torch.enable_grad() if need_nablas else contextlib.nullcontext()):
            d_emb = self.embed_fn_d(d)
            fg_emb = self.embed_fn_fg(interpolation(geometry_features, indices, weights))
            density = self.density_linear(self.pts_linears(torch.cat([d_emb, fg_emb], dim=-1)))
        if not need_nablas:
            return density, torch.zeros_like(density), d_emb
        has_grad = torch.is_grad_enabled()
        nabla = autograd.grad(density, xyz, torch.ones_like(density), create_graph=has_grad, retain_graph=has_grad,
                              only_inputs=True)[0]
        if not has_grad:
            nabla = nabla.detach()
        return density, nabla, d_emb

    def _forward_color(self, d_emb, view_dirs, color_features, indices=None, weights=None, nabla=None):
        """neumesh.py:239-260 in torch ops."""
        parts = [nabla] if self.enable_nablas_input else []
        parts += [d_emb, self.embed_fn_view(view_dirs), self.embed_fn_ft(interpolation(color_features, indices, weights))]
        return self.color_linear(self.views_linears(torch.cat(parts, dim=-1)))
