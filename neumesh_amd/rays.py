"""Ray set-up -- host-side mirror of ``rend_util.get_rays`` (utils/rend_util.py:123-176).

Same signature and return value ``(rays_o, rays_d, select_inds)``.  The case render.py uses (one
pose matrix, ``N_rays=-1``: every pixel in row-major order, render.py:202-207) runs as one HIP
kernel, and ``pixel_range`` lets a rank of a ray-sharded render generate only its own block on its
own GPU.  Random pixel subsets (training) and quaternion poses use torch ops on the device.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _camera(c2w: torch.Tensor, intrinsics: torch.Tensor, H: int, W: int) -> _lib.Camera:
    cam = _lib.Camera()
    m = c2w.detach().float().cpu().reshape(-1)[:12] if c2w.shape[-2:] == (3, 4) else c2w.detach().float().cpu()[:3, :4].reshape(-1)
    for i in range(12):
        cam.c2w[i] = float(m[i])
    k = intrinsics.detach().float().cpu()
    cam.fx, cam.fy, cam.cx, cam.cy, cam.sk = float(k[0, 0]), float(k[1, 1]), float(k[0, 2]), float(k[1, 2]), float(k[0, 1])
    cam.H, cam.W = int(H), int(W)
    return cam


def make_rays(c2w, intrinsics, H, W, device, first_pixel=0, count=None):
    """Rays of pixels [first_pixel, first_pixel+count) -> (rays_o [count,3], rays_d [count,3]) on `device`."""
    lib = _lib.load()
    count = H * W - first_pixel if count is None else count
    cam = _camera(torch.as_tensor(c2w), torch.as_tensor(intrinsics), H, W)
    dev = torch.device(device)
    ro = torch.empty((count, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((count, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.nm_make_rays(C.byref(cam), first_pixel, count, _lib.ptr(ro), _lib.ptr(rd), _lib.current_stream(dev)), "nm_make_rays")
    return ro, rd


def make_rays_indexed(c2w, intrinsics, H, W, pixels: torch.Tensor):
    """Rays of the row-major pixel indices `pixels` (int64 tensor on a HIP device) -> (rays_o [n,3], rays_d [n,3])."""
    lib = _lib.load()
    dev = pixels.device
    if dev.type != "cuda":
        raise _lib.NeuMeshHipError("make_rays_indexed: the pixel list must be on a HIP device (no CPU fallback)")
    pix = pixels.reshape(-1).to(torch.int64).contiguous()
    cam = _camera(torch.as_tensor(c2w), torch.as_tensor(intrinsics), H, W)
    n = pix.shape[0]
    ro = torch.empty((n, 3), dtype=torch.float32, device=dev)
    rd = torch.empty((n, 3), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.nm_make_rays_indexed(C.byref(cam), _lib.ptr(pix), n, _lib.ptr(ro), _lib.ptr(rd), _lib.current_stream(dev)), "nm_make_rays_indexed")
    return ro, rd


def quat_to_rot(q):
    """utils/rend_util.py quaternion (w, x, y, z) -> rotation matrix."""
    q = torch.nn.functional.normalize(q, dim=-1)
    w, x, y, z = q.unbind(-1)
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
        torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
        torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], -2)


def _to_device_async(t: torch.Tensor, device) -> torch.Tensor:
    """Small host tensor -> device without draining the stream: a pageable host-to-device copy is synchronous with respect to the
    host AND ordered in the stream, i.e. it waits for everything queued before it; a copy from pinned memory is just queued."""
    if t.device.type == "cuda":
        return t
    if not torch.cuda.is_available():
        from ._lib import NeuMeshHipError
        raise NeuMeshHipError(f"ray generation for device {device!r}: no HIP device visible (the render / training path has no CPU fallback)")
    return t.pin_memory().to(device, non_blocking=True)


def host_selection(select_inds: torch.Tensor) -> torch.Tensor:
    """Host copy of a pixel list get_rays returned: the fast path draws it on the host and hangs that copy on the very tensor it
    returns (`_nm_host`; no device round trip).  Any other tensor -- the general path's, a caller's own -- is copied back.  (ADVICE r3:
    the copy used to sit in a module global keyed by the device pointer, which the caching allocator can hand to another tensor.)"""
    host = getattr(select_inds, "_nm_host", None)
    if host is not None and host.shape == select_inds.shape:
        return host
    return select_inds.cpu()


def get_rays(c2w, intrinsics, H, W, N_rays=-1, device=None):
    """utils/rend_util.py:123-176.  `device` (extension): where the rays are produced when the pose is handed over on the host --
    the reference moves the pose to the GPU first (trainer.py:61-63); a single pinhole pose is consumed on the host here (the
    camera goes into the kernels by value), so no transfer and no stream synchronisation is needed for it."""
    device = torch.device(device) if device is not None else c2w.device
    if c2w.shape[-1] == 7:  # quaternion + location
        R = quat_to_rot(c2w[..., :4])
        p = torch.eye(4, device=device).repeat([*c2w.shape[:-1], 1, 1]).float()
        p[..., :3, :3] = R
        p[..., :3, 3] = c2w[..., 4:]
    else:
        p = c2w
    prefix = p.shape[:-2]
    single = (p.numel() == 16 and N_rays <= 0 and device.type == "cuda" and intrinsics.numel() == 16)
    if single:
        ro, rd = make_rays(p.reshape(4, 4), intrinsics.reshape(4, 4), H, W, device)
        sel = torch.arange(H * W, device=device).expand([*prefix, H * W])
        return ro.reshape(*prefix, H * W, 3), rd.reshape(*prefix, H * W, 3), sel
    if p.numel() == 16 and N_rays > 0 and device.type == "cuda" and intrinsics.numel() == 16:
        # training batch of one camera: the reference's two host-side draws (same generator, same order => the same pixels),
        # one queued copy of the pixel list, rays by nm_make_rays_indexed
        N_rays = min(N_rays, H * W)
        hs = torch.randint(0, H, size=[N_rays])
        wsel = torch.randint(0, W, size=[N_rays])
        sel_host = hs * W + wsel
        sel = _to_device_async(sel_host, device)
        ro, rd = make_rays_indexed(p.reshape(4, 4), intrinsics.reshape(4, 4), H, W, sel)
        sel_out = sel.expand([*prefix, N_rays])
        sel_out._nm_host = sel_host.expand([*prefix, N_rays])      # (read by host_selection; travels with this tensor object only)
        return ro.reshape(*prefix, N_rays, 3), rd.reshape(*prefix, N_rays, 3), sel_out
    p, intrinsics = p.to(device), intrinsics.to(device)
    # general case in torch ops (same arithmetic as the reference)
    cam_loc = p[..., :3, 3]
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i = i.t().to(device).reshape([*[1] * len(prefix), H * W]).expand([*prefix, H * W])
    j = j.t().to(device).reshape([*[1] * len(prefix), H * W]).expand([*prefix, H * W])
    if N_rays > 0:
        N_rays = min(N_rays, H * W)
        hs = torch.randint(0, H, size=[N_rays]).to(device)
        wsel = torch.randint(0, W, size=[N_rays]).to(device)
        sel = (hs * W + wsel).expand([*prefix, N_rays])
        i, j = torch.gather(i, -1, sel), torch.gather(j, -1, sel)
    else:
        sel = torch.arange(H * W, device=device).expand([*prefix, H * W])
    k = intrinsics.to(device)
    fx, fy, cx, cy, sk = (k[..., 0, 0].unsqueeze(-1), k[..., 1, 1].unsqueeze(-1), k[..., 0, 2].unsqueeze(-1),
                          k[..., 1, 2].unsqueeze(-1), k[..., 0, 1].unsqueeze(-1))
    x_lift = (i - cx + cy * sk / fy - sk * j / fy) / fx
    y_lift = (j - cy) / fy
    d = torch.stack((x_lift, y_lift, torch.ones_like(x_lift)), dim=-1)
    d = d / torch.linalg.norm(d, ord=2, dim=-1, keepdim=True)
    d = torch.matmul(p[..., None, :3, :3], d[..., None]).squeeze(-1)   # rotate every pixel direction by its camera
    return cam_loc[..., None, :].expand_as(d), d, sel
