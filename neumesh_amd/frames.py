"""Frame assembly -- what ``render.py:219-249`` does with the renderer's outputs of one view, kept on the device:

    depth = depth / depth.max();  normals = normals / 2 + 0.5;  img = (x * 255.0).astype(np.uint8)   (``integerify``, :183-184)

``assemble_images`` runs it as one HIP kernel (``nm_assemble_frame``) and returns uint8 device tensors, so a frame
leaves the GPU as 7 bytes per pixel instead of 28; ``write_png`` stores such an array without cv2 / imageio (neither
is a dependency of this package).  SURVEY.md section 8f rank 1.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np
import torch

from . import _lib


def assemble_images(rgb: torch.Tensor, depth: torch.Tensor = None, normals: torch.Tensor = None, H: int = None, W: int = None,
                    bgr: bool = False) -> dict:
    """rgb [..., 3], depth [...], normals [..., 3] of ONE frame (device tensors, H*W pixels in row-major order).
    Returns {"rgb": uint8 [H,W,3], "depth": uint8 [H,W,1], "normal": uint8 [H,W,3]} (device) -- the arrays
    ``render.py`` hands to cv2.imwrite / imageio; bgr=True applies the channel swap of ``render.py:236``."""
    lib = _lib.load()
    rgb = rgb.reshape(-1, 3).float().contiguous()
    n = rgb.shape[0]
    if H is None or W is None:
        H, W = 1, n
    if H * W != n:
        raise ValueError(f"assemble_images: {n} pixels is not {H}x{W}")
    dev = rgb.device
    if dev.type != "cuda":
        raise _lib.NeuMeshHipError("assemble_images: tensors must live on a HIP device; no CPU fallback")
    out = {"rgb": torch.empty((H, W, 3), dtype=torch.uint8, device=dev)}
    d = nrm = None
    scratch = None
    if depth is not None:
        d = depth.reshape(-1).float().contiguous()
        out["depth"] = torch.empty((H, W, 1), dtype=torch.uint8, device=dev)
        scratch = torch.empty((1,), dtype=torch.float32, device=dev)
    if normals is not None:
        nrm = normals.reshape(-1, 3).float().contiguous()
        out["normal"] = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    if (d is not None and d.numel() != n) or (nrm is not None and nrm.shape[0] != n):
        raise ValueError("assemble_images: rgb / depth / normals of different pixel counts")
    with torch.cuda.device(dev):   # (the tensors may live on another device than the current one: sharded / multi-GPU inference)
        _lib.check(lib.nm_assemble_frame(_lib.ptr(rgb), _lib.ptr(d) if d is not None else None, _lib.ptr(nrm) if nrm is not None else None,
                                         n, int(bool(bgr)), _lib.ptr(out["rgb"]), _lib.ptr(out["depth"]) if d is not None else None,
                                         _lib.ptr(out["normal"]) if nrm is not None else None,
                                         _lib.ptr(scratch) if scratch is not None else None, _lib.current_stream(dev)), "nm_assemble_frame")
    return out


def write_png(path: str, img) -> None:
    """uint8 [H,W], [H,W,1] or [H,W,3] (numpy or tensor) -> 8-bit grey / RGB PNG (zlib, no filter)."""
    a = img.detach().cpu().numpy() if torch.is_tensor(img) else np.asarray(img)
    if a.dtype != np.uint8 or a.ndim not in (2, 3) or (a.ndim == 3 and a.shape[2] not in (1, 3)):
        raise ValueError("write_png: expected uint8 [H,W], [H,W,1] or [H,W,3]")
    if a.ndim == 3 and a.shape[2] == 1:
        a = a[..., 0]
    h, w = a.shape[:2]
    rows = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, -1)], axis=1)   # filter type 0 in front of every row

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ihdr = struct.pack(">IIBBBBB", w, h, 8, 2 if a.ndim == 3 else 0, 0, 0, 0)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)) + chunk(b"IEND", b""))
