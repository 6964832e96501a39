"""Ray-sharded multi-GPU rendering: one process per GPU, rays split across ranks, ONE collective
per frame (a gather / all-gather of the final pixels) and nothing else.

The reference has no multi-GPU inference path (its nn.DataParallel / DDP wrap training only,
SURVEY.md section 2 rows 18-19); rays are independent (no cross-ray operation anywhere in
models/renderer.py:162-350), so the mesh index, code tables and MLP weights (~40 MB) are
replicated and rank g renders the contiguous pixel block [g*N/G, (g+1)*N/G).  The per-rank
outputs rgb[3] + depth + acc (+ normals[3]) are packed into one [n, C] fp32 tensor so the
frame costs a single RCCL all-gather of 20-32 B/ray over xGMI (point-to-point links: every peer
is one hop, so no ring is needed).
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch
import torch.distributed as dist

_KEYS = (("rgb", 3), ("depth_volume", 1), ("mask_volume", 1), ("normals_volume", 3))


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block of `n` items owned by `rank` (first n % world ranks get one more)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_outputs(ret: Dict[str, torch.Tensor]) -> Tuple[torch.Tensor, Tuple[str, ...]]:
    keys = tuple(k for k, _ in _KEYS if k in ret)
    cols = [ret[k].reshape(ret[k].shape[0], -1).float() for k in keys]
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


def render_frame_sharded(render_fn, c2w, intrinsics, H: int, W: int, device, group=None) -> Dict[str, torch.Tensor]:
    """One frame of an H x W camera: every rank builds ONLY its own pixel block's rays on its own GPU
    (nm_make_rays, no host->device ray traffic), renders it and all-gathers the pixels."""
    from .rays import make_rays
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = H * W
    lo, hi = shard_range(n, rank, world)
    ro, rd = make_rays(c2w, intrinsics, H, W, device, first_pixel=lo, count=hi - lo)
    ret = render_fn(ro, rd)
    if world == 1:
        return ret
    packed, keys = pack_outputs(ret)
    per = -(-n // world)
    buf = torch.zeros((per, packed.shape[1]), dtype=torch.float32, device=packed.device)
    buf[: hi - lo] = packed
    full = _all_gather_rows(buf, world, group)
    pieces = [full[r * per: r * per + (shard_range(n, r, world)[1] - shard_range(n, r, world)[0])] for r in range(world)]
    return unpack_outputs(torch.cat(pieces, dim=0), keys)
