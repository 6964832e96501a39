"""oracle/cpu_pool.py -- TEST INFRASTRUCTURE ONLY: the CPU oracle on every host core (bench.py's cpu_baseline leg).

`run(npz, workers, rays_per_worker)` starts `workers` fresh interpreters (no fork of a process that holds a GPU context), each
with its math libraries pinned to ONE thread, each rendering its own strided block of the frame's rays with
oracle.render.render_rays (kd-tree K-NN candidates re-ranked with the declared fp32 arithmetic, as in bench.cpu_baseline).
A worker times only its render call (imports and the kd-tree build are outside); the pool's rate is
total rays / the slowest worker's render time, the workers starting within a fraction of a second of each other.
The npz holds: vertices [V,3], rays_o / rays_d [R,3], H, W, samples, white_bkgd, calc_normal, speed_factor and the
model's state dict under 'state/<name>'."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import numpy as np


def _worker(npz_path: str, wid: int, workers: int, rays_per_worker: int) -> None:
    from scipy.spatial import cKDTree
    from oracle import field as ofield, knn as oknn, render as orender
    z = np.load(npz_path)
    state = {k[len("state/"):]: z[k] for k in z.files if k.startswith("state/")}
    verts = z["vertices"]
    orc = ofield.OracleField(verts, state, ofield.FieldConfig(speed_factor=float(z["speed_factor"])))
    tree = cKDTree(verts.astype(np.float64))
    orc.knn_fn = lambda q, v, K: oknn.knn_kdtree(q, v, K, tree=tree)
    o, d = z["rays_o"], z["rays_d"]
    R = o.shape[0]
    sel = (np.linspace(0, R - 1, workers * rays_per_worker).astype(np.int64))[wid::workers]  # this worker's strided block
    samples = int(z["samples"])
    cfg = orender.RenderConfig(calc_normal=bool(z["calc_normal"]), white_bkgd=bool(z["white_bkgd"]), N_samples=samples // 2, N_importance=samples // 2)
    orender.render_rays(orc, o[sel[:4]], d[sel[:4]], cfg)  # warm-up
    t0 = time.time()
    orender.render_rays(orc, o[sel], d[sel], cfg)
    t1 = time.time()
    print(json.dumps({"wid": wid, "rays": int(len(sel)), "t0": t0, "t1": t1}), flush=True)


def run(npz_path: str, workers: int, rays_per_worker: int, timeout: float = 600.0) -> dict:
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS", "VECLIB_MAXIMUM_THREADS"):
        env[k] = "1"
    env["HIP_VISIBLE_DEVICES"] = ""   # workers never touch the GPU
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_pool", npz_path, str(w), str(workers), str(rays_per_worker)],
                              env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for w in range(workers)]
    res = []
    deadline = time.time() + timeout
    for p in procs:
        try:
            out, err = p.communicate(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            for q in procs:
                if q.poll() is None:
                    q.kill()
            raise RuntimeError("cpu_pool: timeout")
        if p.returncode != 0:
            raise RuntimeError("cpu_pool worker failed: " + err[-400:])
        res.append(json.loads(out.strip().splitlines()[-1]))
    rays = sum(r["rays"] for r in res)
    span = max(r["t1"] for r in res) - min(r["t0"] for r in res)      # first render start .. last render end
    slowest = max(r["t1"] - r["t0"] for r in res)
    return {"rays": rays, "seconds": span, "slowest_worker_s": slowest, "rays_per_s": rays / span, "workers": workers}


if __name__ == "__main__":
    _worker(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
