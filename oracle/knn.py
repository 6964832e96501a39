"""oracle/knn.py -- TEST INFRASTRUCTURE ONLY.

Python front end of the declared-arithmetic exact K-NN oracle (``knn_ref.c``), the stand-in
for ``frnn.frnn_grid_points`` as the reference calls it (``models/mesh_grid.py:109-119``:
K=8, r=100.0, return_sorted=True -> squared distances ascending, int64 indices).

Arithmetic: ``dx = q - v`` in IEEE fp32, ``d2 = (dx*dx + dy*dy) + dz*dz``
without FMA contraction, the K smallest by ``(d2, vertex index)`` ascending.

Pinning: the rest of the oracle (field, renderer, editing, training math) is pinned by fixtures generated
from the imported reference (oracle/gen_golden.py).  For THIS boundary only the order of exact ties is a
declaration -- FRNN is external, un-vendored and un-pinned and the reference holds no golden vectors for
it -- so the arithmetic above is checked for self-consistency (C brute force == numpy brute force ==
kd-tree + re-rank) in tests/, and every downstream fixture (which the reference produced through its
kd-tree stand-in with the same re-rank) agrees with it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_knn.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile knn_ref.c with gcc (see oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "knn_ref.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(_LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        lib.nm_oracle_knn.argtypes = [f32p, ctypes.c_int64, f32p, ctypes.c_int64, ctypes.c_int, i64p, f32p]
        lib.nm_oracle_knn.restype = ctypes.c_int
        lib.nm_oracle_rerank.argtypes = [f32p, ctypes.c_int64, f32p, i64p, ctypes.c_int, ctypes.c_int, i64p, f32p]
        lib.nm_oracle_rerank.restype = ctypes.c_int
        _lib = lib
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def knn_bruteforce(q: np.ndarray, verts: np.ndarray, K: int = 8):
    """Exact K-NN, O(Q*V), pinned arithmetic.  Returns (idx int64 [Q,K], d2 f32 [Q,K])."""
    q = _f32(q).reshape(-1, 3)
    verts = _f32(verts).reshape(-1, 3)
    Q, V = q.shape[0], verts.shape[0]
    idx = np.empty((Q, K), dtype=np.int64)
    d2 = np.empty((Q, K), dtype=np.float32)
    lib = _load()
    f32p = ctypes.POINTER(ctypes.c_float)
    i64p = ctypes.POINTER(ctypes.c_int64)
    rc = lib.nm_oracle_knn(q.ctypes.data_as(f32p), Q, verts.ctypes.data_as(f32p), V, K,
                           idx.ctypes.data_as(i64p), d2.ctypes.data_as(f32p))
    if rc != 0:
        raise ValueError("nm_oracle_knn: bad arguments")
    return idx, d2


def knn_numpy(q: np.ndarray, verts: np.ndarray, K: int = 8):
    """Pure-numpy restatement of the same declaration (small cases; cross-checks the C)."""
    q = _f32(q).reshape(-1, 3)
    verts = _f32(verts).reshape(-1, 3)
    dx = q[:, None, 0] - verts[None, :, 0]
    dy = q[:, None, 1] - verts[None, :, 1]
    dz = q[:, None, 2] - verts[None, :, 2]
    d2 = (dx * dx + dy * dy) + dz * dz  # fp32 elementwise, no FMA in numpy
    V = verts.shape[0]
    order = np.lexsort((np.broadcast_to(np.arange(V), d2.shape), d2), axis=-1)[:, :K]
    return order.astype(np.int64), np.take_along_axis(d2, order, axis=1)


def knn_kdtree(q: np.ndarray, verts: np.ndarray, K: int = 8, margin: int = 8, tree=None):
    """kd-tree candidate search (scipy, float64) + re-rank under the pinned fp32 arithmetic.

    Only used to give the CPU *timing* baseline an O(log V) search; the
    parity checks use :func:`knn_bruteforce`.  Equal to it unless fp32 rounding moves a
    vertex across the (K+margin)-th float64 neighbour, which tests/ verify does not
    happen on the benchmark scenes."""
    from scipy.spatial import cKDTree

    q = _f32(q).reshape(-1, 3)
    verts = _f32(verts).reshape(-1, 3)
    if tree is None:
        tree = cKDTree(verts.astype(np.float64))
    C = min(K + margin, verts.shape[0])
    _, cand = tree.query(q.astype(np.float64), k=C, workers=-1)
    cand = np.ascontiguousarray(cand.reshape(q.shape[0], C), dtype=np.int64)
    idx = np.empty((q.shape[0], K), dtype=np.int64)
    d2 = np.empty((q.shape[0], K), dtype=np.float32)
    lib = _load()
    f32p = ctypes.POINTER(ctypes.c_float)
    i64p = ctypes.POINTER(ctypes.c_int64)
    rc = lib.nm_oracle_rerank(q.ctypes.data_as(f32p), q.shape[0], verts.ctypes.data_as(f32p),
                              cand.ctypes.data_as(i64p), C, K,
                              idx.ctypes.data_as(i64p), d2.ctypes.data_as(f32p))
    if rc != 0:
        raise ValueError("nm_oracle_rerank: bad arguments")
    return idx, d2
