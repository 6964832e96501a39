"""tests/hostcheck/loader.py -- TEST INFRASTRUCTURE ONLY.

Builds (g++) and loads libnm_hostcheck.so: the host/device-shared headers of neumesh_amd/csrc
(octree K-NN traversal, projected distance, per-ray stages) compiled for the CPU so their LOGIC
can be checked against the oracle without a GPU.  Never imported by the product."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "neumesh_amd", "csrc")
_SO = os.path.join(_HERE, "libnm_hostcheck.so")
_lib = None
f32p, i64p = C.POINTER(C.c_float), C.POINTER(C.c_int64)


def build(force=False):
    srcs = [os.path.join(_HERE, "hostcheck.cpp")] + [os.path.join(_CSRC, h) for h in
                                                      ("nm_grid.h", "nm_grid_build.h", "nm_distance.h", "nm_rays.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared",
                               "-fPIC", "-I" + _CSRC, srcs[0], "-o", _SO])
    return _SO


def load():
    global _lib
    if _lib is None:
        lib = C.CDLL(build())
        lib.hc_grid_create.restype = C.c_void_p
        lib.hc_grid_create.argtypes = [f32p, C.c_int64, C.c_int]
        lib.hc_grid_destroy.argtypes = [C.c_void_p]
        lib.hc_grid_level.argtypes = [C.c_void_p]
        lib.hc_grid_occupied.argtypes = [C.c_void_p]
        lib.hc_knn.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int, i64p, f32p]
        lib.hc_knn_stats.argtypes = [C.c_void_p, f32p, C.c_int64, C.POINTER(C.c_double)]
        lib.hc_knn_warm.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, i64p, f32p, C.POINTER(C.c_double)]
        lib.hc_ray_upsample_slots.argtypes = [f32p, f32p, C.POINTER(C.c_int32), f32p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.hc_knn_packet.argtypes = [C.c_void_p, f32p, C.c_int64, C.c_int, i64p, f32p, C.POINTER(C.c_double)]
        lib.hc_compute_distance.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, C.c_float, f32p, i64p, f32p, f32p]
        lib.hc_linspace01.argtypes = [C.c_int, f32p]
        lib.hc_ray_setup.argtypes = [f32p, f32p, C.c_int64, C.c_float, f32p, f32p]
        lib.hc_ray_bounds.argtypes = [f32p, C.c_int64, C.c_int, C.c_float, f32p, f32p]
        lib.hc_ray_upsample.argtypes = [f32p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.hc_ray_upsample_u.argtypes = [f32p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        lib.hc_ray_merge.argtypes = [f32p, f32p, C.c_int64, C.c_int, C.c_int, C.c_int]
        lib.hc_ray_composite.argtypes = [f32p, f32p, C.c_int64, C.c_int, C.c_float, f32p, f32p, C.c_int, f32p, f32p, f32p, f32p]
        _lib = lib
    return _lib


def P(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(f32p if a.dtype == np.float32 else i64p)


class HostGrid:
    def __init__(self, verts, leaf_level=0):
        self.lib = load()
        self.verts = np.ascontiguousarray(verts, np.float32)
        self.h = self.lib.hc_grid_create(P(self.verts), len(self.verts), leaf_level)
        assert self.h, "hc_grid_create failed"

    @property
    def level(self):
        return self.lib.hc_grid_level(self.h)

    def knn(self, q, K):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        idx = np.empty((len(q), K), np.int64)
        d2 = np.empty((len(q), K), np.float32)
        assert self.lib.hc_knn(self.h, P(q), len(q), K, P(idx), P(d2)) == 0
        return idx, d2

    def knn_stats(self, q):
        """(node records tested, vertices scanned) per query, averaged."""
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        out = (C.c_double * 2)()
        self.lib.hc_knn_stats(self.h, P(q), len(q), out)
        return out[0], out[1]

    def knn_warm(self, q, bound):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(bound, np.float32).reshape(-1)
        idx = np.empty((len(q), 8), np.int64)
        d2 = np.empty((len(q), 8), np.float32)
        out = (C.c_double * 2)()
        assert self.lib.hc_knn_warm(self.h, P(q), len(q), P(b), P(idx), P(d2), out) == 0
        return idx, d2, out[0], out[1]

    def knn_packet(self, q, width=64):
        """Packet traversal emulation: (idx, d2, nodes per packet, vertices per packet)."""
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        idx = np.empty((len(q), 8), np.int64)
        d2 = np.empty((len(q), 8), np.float32)
        out = (C.c_double * 9)()
        assert self.lib.hc_knn_packet(self.h, P(q), len(q), width, P(idx), P(d2), out) == 0
        self.last_packet_stats = {"insert_events_per_packet": out[2], "deferred_rounds_per_packet": out[3],
                                  "inserts_per_query": out[4], "marked_per_query": out[5], "node_tests_no_lane_passed_per_packet": out[6],
                                  "caught_by_axis_separating_packet_test": out[7], "caught_by_packet_box_bound": out[8]}
        return idx, d2, out[0], out[1]

    def compute_distance(self, q, indicator, w1):
        q = np.ascontiguousarray(q, np.float32).reshape(-1, 3)
        ind = np.ascontiguousarray(indicator, np.float32)
        n = len(q)
        ds, idx, w, g = np.empty(n, np.float32), np.empty((n, 8), np.int64), np.empty((n, 8), np.float32), np.empty((n, 3), np.float32)
        assert self.lib.hc_compute_distance(self.h, P(q), n, P(ind), w1, P(ds), P(idx), P(w), P(g)) == 0
        return ds, idx, w, g

    def __del__(self):
        try:
            self.lib.hc_grid_destroy(self.h)
        except Exception:
            pass
