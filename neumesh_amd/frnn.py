"""Drop-in for the one function NeuMesh uses from the external FRNN CUDA package:

    frnn.frnn_grid_points(points1, points2, lengths1, lengths2, K, r, grid, return_nn, return_sorted)
        -> (dists [1,Q,K] f32 squared, idxs [1,Q,K] int64, nn (None), grid)

(call sites: reference models/mesh_grid.py:64-74 and :109-119).  A maintainer who wants to keep
the reference's own mesh_grid.py can replace ``import frnn`` by
``from neumesh_amd import frnn`` and nothing else.  Results follow the
declared arithmetic of include/neumesh_hip.h; neighbours farther than ``r`` are padded with -1
exactly like FRNN (never the case at the reference's r=100 in a unit-sphere scene)."""
from __future__ import annotations

import torch

from .mesh_grid import GridHandle, knn


def frnn_grid_points(points1, points2, lengths1=None, lengths2=None, K=-1, r=-1.0, grid=None,
                     return_nn=False, return_sorted=True, radius_cell_ratio=2.0):
    if points1.dim() != 3 or points2.dim() != 3 or points1.shape[0] != 1 or points2.shape[0] != 1:
        raise ValueError("frnn_grid_points: only batch size 1 is supported (what NeuMesh uses)")
    if lengths1 is not None or lengths2 is not None:
        raise ValueError("frnn_grid_points: ragged lengths are not supported")
    if return_nn:
        raise ValueError("frnn_grid_points: return_nn=True is not supported")
    if not (1 <= K <= 32):
        raise ValueError("frnn_grid_points: K must be in [1,32]")
    if grid is None:
        grid = GridHandle(points2[0])
    elif not isinstance(grid, GridHandle):
        raise TypeError("frnn_grid_points: `grid` must be the object returned by a previous call")
    idx, d2 = knn(grid, points1[0], K)
    if r is not None and r > 0:
        far = d2 > float(r) * float(r)  # no host sync: plain element-wise selects
        idx = torch.where(far, torch.full_like(idx, -1), idx)
        d2 = torch.where(far, torch.full_like(d2, -1.0), d2)
    return d2.unsqueeze(0), idx.unsqueeze(0), None, grid
