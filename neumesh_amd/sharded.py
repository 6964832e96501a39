"""Ray-sharded multi-GPU rendering: one process per GPU, rays split across ranks, ONE collective
per frame (a gather / all-gather of the final pixels) and nothing else.

The reference has no multi-GPU inference path (its nn.DataParallel / DDP wrap training only,
SURVEY.md section 2 rows 18-19); rays are independent (no cross-ray operation anywhere in
models/renderer.py:162-350), so the mesh index, code tables and MLP weights (~40 MB) are
replicated.  A frame is cut into TILE x TILE pixel tiles dealt round-robin to the ranks
(`tile_shard_pixels`): the cost of a ray depends on where it goes -- rays that miss the object walk
all 256 near/far probes and then carry no mid-point work, rays through the surface the opposite --
so contiguous bands of an image are unevenly loaded while every rank's share of interleaved 32 x 32
tiles sees the same mix.  (`render_sharded`, for caller-supplied ray lists whose layout is unknown,
keeps contiguous blocks.)  The per-rank outputs rgb[3] + depth + acc (+ normals[3]) are packed into one
[n, C] fp32 tensor so the frame costs a single RCCL all-gather of 20-32 B/ray over xGMI (point-to-point
links: every peer is one hop, so no ring is needed); every rank then scatters the rows to pixel order
with the tile table it can compute by itself.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch
import torch.distributed as dist

TILE = 32   # pixels per tile edge: 625 tiles of an 800 x 800 frame, 1900 of 1600 x 1200

_KEYS = (("rgb", 3), ("depth_volume", 1), ("mask_volume", 1), ("normals_volume", 3))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of `n` items owned by `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_outputs(ret: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, Tuple[str, ...]]:
    keys = tuple(k for k, _ in _KEYS if k in ret)
    cols = [ret[k].reshape(ret[k].shape[0], w).float() for k, w in _KEYS if k in ret]   # (explicit widths: a rank may hold 0 rays)
    return torch.cat(cols, dim=1).contiguous(), keys


def unpack_outputs(packed: torch.Tensor, keys: Tuple[str, ...]) -> Dict[str, torch.Tensor]:
    out, c = {}, 0
    for k, w in _KEYS:
        if k in keys:
            v = packed[:, c:c + w]
            out[k] = v if w == 3 else v[:, 0]
            c += w
    return out


def _all_gather_rows(buf: torch.Tensor, world: int, group=None) -> torch.Tensor:
    """[per, C] per rank -> [world*per, C] on every rank: ONE collective.  RCCL ("nccl") gathers device
    tensors directly over xGMI; under a host backend (gloo: CPU-only test runs, or two ranks sharing one
    GPU in the single-GPU parity test) the rows are staged through host memory."""
    full = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype, device=buf.device)
    if buf.is_cuda and dist.get_backend(group) != "nccl":
        host = torch.empty(full.shape, dtype=buf.dtype)
        dist.all_gather_into_tensor(host, buf.cpu(), group=group)
        full.copy_(host)
    else:
        dist.all_gather_into_tensor(full, buf, group=group)
    return full


class PendingGather:
    """An all-gather of pixel rows that has been POSTED but not waited for (round 6: frame i's collective runs on the back end's own
    stream while frame i + 1 renders -- the renderer no longer synchronises, renderer.py).  result() makes the CURRENT stream wait for the
    collective (no host synchronisation under RCCL) and returns the gathered [world * per, C] rows."""

    def __init__(self, work, full, host=None, keep=None):
        self._work, self._full, self._host, self._keep = work, full, host, keep

    def result(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
            if self._host is not None:      # host back end: the rows were gathered in host memory
                self._full.copy_(self._host)
                self._host = None
            self._keep = None
        return self._full


def all_gather_rows_async(buf: torch.Tensor, world: int, group=None) -> PendingGather:
    """_all_gather_rows with async_op=True: the collective is queued behind everything already on the current stream and the call returns;
    kernels queued on the current stream AFTERWARDS (the next frame) do not wait for it."""
    full = torch.empty((world * buf.shape[0], buf.shape[1]), dtype=buf.dtype, device=buf.device)
    if buf.is_cuda and dist.get_backend(group) != "nccl":
        host, src = torch.empty(full.shape, dtype=buf.dtype), buf.cpu()
        return PendingGather(dist.all_gather_into_tensor(host, src, group=group, async_op=True), full, host, src)
    return PendingGather(dist.all_gather_into_tensor(full, buf, group=group, async_op=True), full, None, buf)


def render_sharded(render_fn: Callable[[torch.Tensor, torch.Tensor], Dict[str, torch.Tensor]], rays_o: torch.Tensor,
                   rays_d: torch.Tensor, group=None) -> Dict[str, torch.Tensor]:
    """Every rank passes the SAME full [N,3] rays (or its rank could build them on device, rays
    are cheap); each renders its block with `render_fn(rays_o_block, rays_d_block) -> dict` and
    all ranks return the full-frame dict after one all-gather."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return render_fn(rays_o, rays_d)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = rays_o.shape[0]
    lo, hi = shard_range(n, rank, world)
    packed, keys = pack_outputs(render_fn(rays_o[lo:hi], rays_d[lo:hi]))
    width = packed.shape[1]
    per = -(-n // world)  # padded shard size so that the collective is a plain all_gather
    buf = torch.zeros((per, width), dtype=torch.float32, device=packed.device)
    buf[: hi - lo] = packed
    full = _all_gather_rows(buf, world, group)
    pieces = []
    for r in range(world):
        a, b = shard_range(n, r, world)
        pieces.append(full[r * per: r * per + (b - a)])
    return unpack_outputs(torch.cat(pieces, dim=0), keys)


def tile_shard_pixels(H: int, W: int, rank: int, world: int, tile: int = TILE, device=None) -> torch.Tensor:
    """Row-major pixel indices (int64) of the tiles rank `rank` owns: tiles are numbered row-major over the
    ceil(H/tile) x ceil(W/tile) tile grid, tile t belongs to rank t % world; inside a tile pixels run row-major
    (partial tiles at the right / bottom border simply hold fewer pixels).  The `world` lists partition range(H*W)."""
    tiles_x, tiles_y = -(-W // tile), -(-H // tile)
    if rank >= tiles_x * tiles_y:
        return torch.empty((0,), dtype=torch.int64, device=device)
    t = torch.arange(rank, tiles_x * tiles_y, world, device=device, dtype=torch.int64)
    ty, tx = t // tiles_x, t % tiles_x
    iy = torch.arange(tile, device=device, dtype=torch.int64)
    py = (ty[:, None, None] * tile + iy[None, :, None]).expand(-1, tile, tile)
    px = (tx[:, None, None] * tile + iy[None, None, :]).expand(-1, tile, tile)
    ok = (py < H) & (px < W)
    return (py * W + px)[ok]


def pick_tile(H: int, W: int, world: int, tile: int = TILE) -> int:
    """Tile edge for an H x W frame over `world` ranks: TILE, halved (down to 8 = one 64-ray wave's patch) until every rank
    gets at least 16 tiles -- small frames would otherwise be dealt out unevenly."""
    while tile > 8 and (-(-H // tile)) * (-(-W // tile)) < 16 * world:
        tile //= 2
    return tile


_TABLES = {}


def _frame_tables(H, W, world, tile, device):
    """(per-rank pixel lists, padded rows per rank, scatter index of the gathered [world*per] rows -> pixel order)."""
    key = (H, W, world, tile, str(device))
    tab = _TABLES.get(key)
    if tab is None:
        lists = [tile_shard_pixels(H, W, r, world, tile, device) for r in range(world)]
        per = max(int(p.shape[0]) for p in lists)
        rows = torch.cat([r * per + torch.arange(p.shape[0], device=device, dtype=torch.int64) for r, p in enumerate(lists)])
        pix = torch.cat(lists)
        src = torch.empty(H * W, dtype=torch.int64, device=device)
        src[pix] = rows                      # pixel p of the frame is row src[p] of the gathered buffer
        if len(_TABLES) > 8:
            _TABLES.clear()
        tab = _TABLES[key] = (lists, per, src)
    return tab


def render_frame_sharded(render_fn, c2w, intrinsics, H: int, W: int, device, group=None, tile: int = None,
                         timings: dict = None) -> Dict[str, torch.Tensor]:
    """One frame of an H x W camera over the ranks of `group`: every rank builds ONLY the rays of its own interleaved
    tiles on its own GPU (nm_make_rays_indexed, no host->device ray traffic), renders them with
    `render_fn(rays_o, rays_d) -> dict` and all ranks end with the full frame in pixel order after ONE all-gather.
    timings (optional dict): receives "render_ms" (this rank's render, host clock around a device sync) when given."""
    from .rays import make_rays, make_rays_indexed
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        ro, rd = make_rays(c2w, intrinsics, H, W, device)
        return render_fn(ro, rd)
    lists, per, src = _frame_tables(H, W, world, tile or pick_tile(H, W, world), torch.device(device))
    mine = lists[rank]
    ro, rd = make_rays_indexed(c2w, intrinsics, H, W, mine)
    if timings is not None:
        import time
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
    ret = render_fn(ro, rd)
    if timings is not None:
        torch.cuda.synchronize(device)
        timings["render_ms"] = (time.perf_counter() - t0) * 1e3
        timings["rays"] = int(mine.shape[0])
    return gather_tiles(ret, per, src, world, group)


def render_frame_sharded_async(render_fn, c2w, intrinsics, H: int, W: int, device, group=None, tile: int = None):
    """render_frame_sharded with the frame's all-gather only POSTED: returns a function that waits for the collective and assembles the
    frame.  Call it after the NEXT frame's render has been queued and the two overlap (render_frames_sharded does exactly that)."""
    from .rays import make_rays, make_rays_indexed
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        ro, rd = make_rays(c2w, intrinsics, H, W, device)
        ret = render_fn(ro, rd)
        return lambda: ret
    lists, per, src = _frame_tables(H, W, world, tile or pick_tile(H, W, world), torch.device(device))
    ro, rd = make_rays_indexed(c2w, intrinsics, H, W, lists[rank])
    return gather_tiles_async(render_fn(ro, rd), per, src, world, group)


def gather_tiles(ret: Dict[str, torch.Tensor], per: int, src: torch.Tensor, world: int, group=None) -> Dict[str, torch.Tensor]:
    """This rank's per-ray outputs (in the order of its tile_shard_pixels list) -> the full frame in pixel order on
    every rank: rows padded to `per`, ONE all-gather, one index_select with the `src` table of _frame_tables."""
    return gather_tiles_async(ret, per, src, world, group)()


def gather_tiles_async(ret: Dict[str, torch.Tensor], per: int, src: torch.Tensor, world: int, group=None):
    """gather_tiles with the collective only POSTED: returns a function that waits for it and assembles the frame."""
    packed, keys = pack_outputs(ret)
    buf = torch.zeros((per, packed.shape[1]), dtype=torch.float32, device=packed.device)
    buf[: packed.shape[0]] = packed
    pending = all_gather_rows_async(buf, world, group)
    return lambda: unpack_outputs(pending.result()[src], keys)


def render_frames_sharded(render_fn, cameras, H: int, W: int, device, group=None, tile: int = None):
    """A SEQUENCE of frames (render.py:294's 90-view spiral) over the ranks of `group`, pipelined: yields the full frame of camera i
    (pixel order, on every rank) after camera i + 1's render has been queued, so frame i's all-gather -- on the back end's stream -- and its
    assembly overlap the next frame's kernels instead of standing between two frames.  cameras: iterable of (c2w, intrinsics).
    Same pixels as render_frame_sharded frame by frame."""
    waiting = None
    for c2w, intrinsics in cameras:
        nxt = render_frame_sharded_async(render_fn, c2w, intrinsics, H, W, device, group, tile)
        if waiting is not None:
            yield waiting()
        waiting = nxt
    if waiting is not None:
        yield waiting()
