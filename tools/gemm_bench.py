"""tools/gemm_bench.py -- GPU box: the training path's GEMM alone (csrc/nm_gemm.h through the testing library's nm_debug_gemm): launch time
and error against float64 for the three products of a linear layer, fp32 pipe vs bf16 x 3.

usage: python tools/gemm_bench.py [rows=65536]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from neumesh_amd import _lib
dev = torch.device("cuda", 0)
lib = _lib.load_testing()
st = _lib.current_stream(dev)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
W = 256
g = torch.Generator(device="cpu").manual_seed(1)
X = torch.randn(P, W, generator=g).to(dev)
Wt = (torch.randn(W, W, generator=g) * 0.06).to(dev)
dY = (torch.randn(P, W, generator=g) * 1e-6).to(dev)      # cotangent-sized values


def run(A, lda, akc, B, ldb, bkc, M, N, K, split, mode, iters=20):
    Cc = torch.zeros(M, N, device=dev)
    ms = C.c_float(0)
    _lib.check(lib.nm_debug_gemm(_lib.ptr(A), lda, akc, _lib.ptr(B), ldb, bkc, _lib.ptr(Cc), N, M, N, K, None, 0, split, 1 if split > 1 else 0, mode, 1, C.byref(ms), st), "gemm")
    out = Cc.clone()
    _lib.check(lib.nm_debug_gemm(_lib.ptr(A), lda, akc, _lib.ptr(B), ldb, bkc, _lib.ptr(Cc), N, M, N, K, None, 0, split, 1 if split > 1 else 0, mode, iters, C.byref(ms), st), "gemm")
    return out, ms.value


cases = [("forward      Y = X W^T", X, W, 1, Wt, W, 1, P, W, W, 1, lambda: X.double() @ Wt.double().T),
         ("input grad   dX = dY W", dY, W, 1, Wt, W, 0, P, W, W, 1, lambda: dY.double() @ Wt.double()),
         ("weight grad  dW = dY^T X", dY, W, 0, X, W, 0, W, W, P, max(1, min(256, P // 512)), lambda: dY.double().T @ X.double())]
for name, A, lda, akc, B, ldb, bkc, M, N, K, split, ref in cases:
    r = ref()
    scale = float(r.abs().max())
    flops = 2.0 * M * N * K
    byts = 4.0 * (M * K + K * N + M * N)
    for mode, tag in ((0, "fp32 pipe"), (1, "bf16 x 3 ")):
        out, ms = run(A, lda, akc, B, ldb, bkc, M, N, K, split, mode)
        err = float((out.double() - r).abs().max()) / scale
        print(f"{name:26s} {tag}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s  {byts / ms / 1e9:6.2f} TB/s   max err / max |C| = {err:.2e}  (M {M}, N {N}, K {K}, split {split})", flush=True)
