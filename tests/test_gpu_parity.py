"""GPU (-m gpu): the HIP path, called through the C ABI (ctypes), against the golden fixtures
(= the reference's own outputs) and the oracle, plus size-independent properties at the
BASELINE.json sizes.  Tolerances: K-NN indices / squared distances bit-exact; everything
floating point within the 1e-4 RGB bound of BASELINE.json:north_star (most stages are 1e-6)."""
import ctypes as C
import os

import numpy as np
import pytest

import common
from oracle import compare, knn as oknn, render as orender

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    return torch


@pytest.fixture(scope="module")
def small(cuda_device):
    mesh = common.scene_mesh(3000)
    state = common.scene_state(mesh)
    return mesh, state, common.make_model(mesh, state, cuda_device)


@pytest.fixture(scope="module")
def dtu_scale(cuda_device):
    mesh = common.scene_mesh(140000)
    state = common.scene_state(mesh)
    return mesh, state, common.make_model(mesh, state, cuda_device)


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_native_library_is_the_one_in_tree(cuda_device):
    from neumesh_amd import _lib, build
    lib = _lib.load()
    assert lib.nm_device_count() >= 1
    assert build.LIB_PATH.endswith("neumesh_amd/csrc/libneumesh_hip.so")
    assert any("libneumesh_hip.so" in line for line in open("/proc/self/maps"))


# --------------------------------------------------------------------------------- K-NN
@pytest.mark.parametrize("K", [1, 3, 8, 16, 32])
def test_knn_bit_exact_vs_oracle(small, cuda_device, K):
    from neumesh_amd.mesh_grid import knn
    mesh, _, model = small
    fx = common.golden("field_v3000")
    idx, d2 = knn(model.mesh_grid.grid, _t(fx["q"], cuda_device), K)
    ridx, rd2 = oknn.knn_bruteforce(fx["q"], mesh.vertices, K)
    assert idx.dtype == np.int64 or str(idx.dtype) == "torch.int64"
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert np.array_equal(d2.cpu().numpy(), rd2)
    if K == 8:
        assert np.array_equal(idx.cpu().numpy(), fx["idx"]) and np.array_equal(d2.cpu().numpy(), fx["d2"])


def test_knn_duplicate_vertices_ties_by_index(cuda_device):
    from neumesh_amd.mesh_grid import knn
    fx = common.golden("field_dup_v1200")
    mesh = common.scene_mesh(1200, 64)
    model = common.make_model(mesh, common.scene_state(mesh), cuda_device)
    idx, d2 = knn(model.mesh_grid.grid, _t(fx["q"], cuda_device), 8)
    assert np.array_equal(idx.cpu().numpy(), fx["idx"]) and np.array_equal(d2.cpu().numpy(), fx["d2"])


def test_knn_fewer_vertices_than_K_pads_minus_one(cuda_device, torch_mod):
    from neumesh_amd import frnn
    torch = torch_mod
    v = torch.randn(1, 5, 3, device=cuda_device)
    q = torch.randn(1, 64, 3, device=cuda_device)
    d2, idx, _, grid = frnn.frnn_grid_points(q, v, None, None, K=8, r=100.0, grid=None, return_nn=False, return_sorted=True)
    ridx, rd2 = oknn.knn_bruteforce(q[0].cpu().numpy(), v[0].cpu().numpy(), 8)
    assert np.array_equal(idx[0].cpu().numpy(), ridx) and np.array_equal(d2[0].cpu().numpy(), rd2)
    assert (idx[0, :, 5:] == -1).all()
    d2b, idxb, _, grid2 = frnn.frnn_grid_points(q, v, None, None, K=4, r=100.0, grid=grid)   # cached grid, like mesh_grid.py:109-119
    assert grid2 is grid and np.array_equal(idxb[0].cpu().numpy(), ridx[:, :4])
    d2c, idxc, _, _ = frnn.frnn_grid_points(q, v, None, None, K=4, r=0.5, grid=grid)         # finite radius -> -1 padding
    assert ((d2c[0] > 0.25) == False).all() and ((idxc[0] == -1) == (d2c[0] == -1)).all()  # noqa: E712


def test_knn_empty_and_ragged_sizes(small, cuda_device, torch_mod):
    from neumesh_amd.mesh_grid import knn
    mesh, _, model = small
    idx, d2 = knn(model.mesh_grid.grid, torch_mod.zeros((0, 3), device=cuda_device), 8)
    assert idx.shape == (0, 8)
    rng = np.random.default_rng(5)
    for Q in (1, 63, 257, 1000):
        q = rng.uniform(-1, 1, (Q, 3)).astype(np.float32)
        idx, d2 = knn(model.mesh_grid.grid, _t(q, cuda_device), 8)
        ridx, rd2 = oknn.knn_bruteforce(q, mesh.vertices, 8)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)


def test_knn_dtu_scale_vs_oracle_and_properties(dtu_scale, cuda_device, torch_mod):
    """V = 1.4e5 (BASELINE.json config 2 shape): bit-exact vs brute force on 20 k queries of every
    regime; on 2 M queries the size-independent properties (sorted, consistent, deterministic)."""
    from neumesh_amd.mesh_grid import knn
    torch = torch_mod
    mesh, _, model = dtu_scale
    rng = np.random.default_rng(9)
    V = mesh.num_vertices
    q = np.concatenate([mesh.vertices[rng.integers(0, V, 10000)] + 0.01 * rng.standard_normal((10000, 3)),
                        mesh.vertices[rng.integers(0, V, 5000)] + 0.2 * rng.standard_normal((5000, 3)),
                        rng.uniform(-2, 2, (4990, 3)), mesh.vertices[:10]]).astype(np.float32)
    idx, d2 = knn(model.mesh_grid.grid, _t(q, cuda_device), 8)
    ridx, rd2 = oknn.knn_bruteforce(q, mesh.vertices, 8)
    assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)
    Q = 1 << 21
    g = torch.Generator(device="cpu").manual_seed(1)
    big = (torch.from_numpy(mesh.vertices)[torch.randint(0, V, (Q,), generator=g)] + 0.05 * torch.randn(Q, 3, generator=g)).to(cuda_device)
    idx, d2 = knn(model.mesh_grid.grid, big, 8)
    idx2, d22 = knn(model.mesh_grid.grid, big, 8)
    assert torch.equal(idx, idx2) and torch.equal(d2, d22)                       # deterministic
    assert bool((d2[:, 1:] >= d2[:, :-1]).all())                                 # ascending
    ties = d2[:, 1:] == d2[:, :-1]
    assert bool((idx[:, 1:][ties] > idx[:, :-1][ties]).all())                   # ties by index
    verts = torch.from_numpy(mesh.vertices).to(cuda_device)
    diff = big[:, None, :] - verts[idx]
    rec = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert torch.equal(rec, d2)                                                  # d2 is the declared arithmetic
    assert int(idx.min()) >= 0 and int(idx.max()) < V
    assert bool((idx.sort(dim=1).values[:, 1:] != idx.sort(dim=1).values[:, :-1]).all())  # 8 distinct vertices


def test_knn_cooperative_and_private_paths_agree(dtu_scale, cuda_device, torch_mod):
    """The wave-cooperative traversal (compact 64-query groups: consecutive samples along rays,
    clustered points) and the lane-private one (scattered points) both equal brute force."""
    from neumesh_amd import synthetic
    from neumesh_amd.mesh_grid import knn
    mesh, _, model = dtu_scale
    o, d = synthetic.camera_rays(synthetic.orbit_pose(2), synthetic.pinhole_intrinsics(800, 800), 800, 800, start=400 * 800 + 300, count=48)
    d = orender.normalize(d)
    t = np.linspace(1.2, 3.2, 200, dtype=np.float32)
    coherent = (o[:, None, :] + t[None, :, None] * d[:, None, :]).reshape(-1, 3).astype(np.float32)   # ray-major: compact waves
    rng = np.random.default_rng(3)
    scattered = coherent[rng.permutation(len(coherent))]                                              # same points, shuffled
    clustered = (mesh.vertices[777] + 1e-3 * rng.standard_normal((640, 3))).astype(np.float32)
    for pts in (coherent, scattered, clustered):
        idx, d2 = knn(model.mesh_grid.grid, _t(pts, cuda_device), 8)
        ridx, rd2 = oknn.knn_bruteforce(pts, mesh.vertices, 8)
        assert np.array_equal(idx.cpu().numpy(), ridx) and np.array_equal(d2.cpu().numpy(), rd2)


def test_small_launch_deferral_is_exact(dtu_scale, cuda_device, torch_mod, monkeypatch):
    """Small launches (<= 2^18 points) give up a wave's traversal after a work budget and finish its queries subtree by subtree on the
    whole chip (csrc/nm_kernels.h "the deferred queries of a small launch"; budget: NEUMESH_KNN_BUDGET, 0 = off, 1 = defer whatever
    the list has room for).  Every output of the fused K-NN + distance kernel is identical bit for bit whichever way a query went:
    points near the centre of the object (the expensive ones: every surface patch about equally far), ray-ordered probes, scattered
    points; 20 000 points, so that budget 1 also overflows the 8192-entry list (the waves that find it full finish on their own), and
    a second call right after with fewer points (entries of the earlier launch must not be picked up)."""
    torch = torch_mod
    from neumesh_amd import synthetic
    mesh, _, model = dtu_scale
    rng = np.random.default_rng(21)
    o, d = synthetic.camera_rays(synthetic.orbit_pose(4), synthetic.pinhole_intrinsics(800, 800), 800, 800, start=400 * 800 + 380, count=40)
    d = orender.normalize(d)
    t = np.linspace(1.0, 3.4, 256, dtype=np.float32)
    probes = (o[:, None, :] + t[None, :, None] * d[:, None, :]).reshape(-1, 3)
    centre = 0.05 * rng.standard_normal((4000, 3))
    scattered = rng.uniform(-1.2, 1.2, (5760, 3))
    q = _t(np.concatenate([probes, centre, scattered]).astype(np.float32), cuda_device)
    assert q.shape[0] == 20000
    ind = model.indicator_vector.detach()

    def run(points):
        with torch.no_grad():
            ds, idx, w = model.compute_distance(points)
            feat = model.mesh_grid.compute_distance_interpolate(points, model.geometry_features.detach(), ind, 0.1)
        return [ds, idx, w] + list(feat)

    monkeypatch.setenv("NEUMESH_KNN_BUDGET", "0")
    ref, ref_small = run(q), run(q[:3000])
    ridx, _ = oknn.knn_bruteforce(q[:2000].cpu().numpy(), mesh.vertices, 8)
    assert np.array_equal(ref[1][:2000].cpu().numpy(), ridx)
    for budget in ("1", "3000", "30000"):
        monkeypatch.setenv("NEUMESH_KNN_BUDGET", budget)
        for got, want in ((run(q), ref), (run(q[:3000]), ref_small)):
            assert len(got) == len(want)
            for a, b in zip(got, want):
                assert torch.equal(a, b), budget
    monkeypatch.delenv("NEUMESH_KNN_BUDGET")
    for a, b in zip(run(q), ref):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------- field
def test_compute_distance_matches_reference(small, cuda_device, torch_mod):
    _, _, model = small
    fx = common.golden("field_v3000")
    q = _t(fx["q"], cuda_device)
    with torch_mod.no_grad():
        ds, idx, w = model.compute_distance(q)
        ds2, _, _, g = model.mesh_grid.compute_distance_frnn(q, 8, model.indicator_vector, 0.1, want_grad=True)
    assert tuple(ds.shape) == (len(fx["q"]), 1) and str(idx.dtype) == "torch.int64"
    assert np.array_equal(idx.cpu().numpy(), fx["idx"])
    np.testing.assert_allclose(ds.cpu().numpy(), fx["ds"], atol=3e-6)
    np.testing.assert_allclose(w.cpu().numpy(), fx["w"], atol=1e-6)
    np.testing.assert_allclose(g.cpu().numpy(), fx["dds_dx"], atol=2e-5, rtol=2e-5)
    assert torch_mod.equal(ds, ds2)


@pytest.mark.parametrize("precision", ["f16x2s", "f16x2", "fp32"])
def test_field_methods_match_reference(small, cuda_device, torch_mod, precision):
    torch = torch_mod
    _, _, model = small
    model.mlp_precision = precision     # both MLP arithmetic modes must meet the same bars
    fx = common.golden("field_v3000")
    q, dirs = _t(fx["q"], cuda_device), _t(fx["dirs"], cuda_device)
    tol_nab = 5e-6 + 2e-4 * np.abs(fx["ds"])   # fp32 sensitivity of the 2^7-band d-embedding, see oracle/gen_golden.py
    with torch.no_grad():
        sdf = model.forward_density_only(q)
        sdf2, nab = model.forward_with_nablas(q)
        sdf3, rgb, ds, idx, w = model.forward(q, dirs, return_ds=True)
        sdf4, nab2 = model.forward(q, dirs, nablas_only=True)
        rgb2 = model.forward_color(ds, dirs, model.color_features, idx, w, nab)
    np.testing.assert_allclose(sdf.cpu().numpy(), fx["sdf"], atol=3e-6)
    assert torch.equal(sdf, sdf2) and torch.equal(sdf, sdf3) and torch.equal(sdf, sdf4)   # one value per point, whatever the path
    assert np.all(np.abs(nab.cpu().numpy() - fx["nabla"]) <= tol_nab)
    assert torch.equal(nab, nab2)
    np.testing.assert_allclose(rgb.cpu().numpy(), fx["rgb"], atol=3e-6)
    assert torch.equal(rgb, rgb2)
    assert np.array_equal(idx.cpu().numpy(), fx["idx"])
    assert abs(float(model.forward_s()) - float(fx["s"])) < 1e-3
    model.mlp_precision = common.DEFAULT_PRECISION


def test_mfma_tile_code_equals_scalar_alu_reference(small, cuda_device, torch_mod):
    """Device cross-check of the MFMA fragment layout: same kernel with the layer computed by a
    plain per-thread fmaf loop (asymmetric weights/activations => a transposed tile would fail)."""
    torch = torch_mod
    from neumesh_amd import _lib
    lib = _lib.load_testing()         # nm_selfcheck_field is a test hook: only the -DNM_TESTING build of the library has it
    _, _, model = small               # (the handles are plain structs of device pointers: the product library's work here too)
    model.mlp_precision = "fp32"      # this test is about the fp32 MFMA tile code
    fx = common.golden("field_v3000")
    q, dirs = _t(fx["q"], cuda_device), _t(fx["dirs"], cuda_device)
    P = q.shape[0]
    sdf, nab, rgb = torch.empty(P, device=cuda_device), torch.empty(P, 3, device=cuda_device), torch.empty(P, 3, device=cuda_device)
    scratch = torch.empty(int(lib.nm_field_scratch_bytes(P)), dtype=torch.uint8, device=cuda_device)
    tmp = torch.empty((P + 31) // 32 * 64 * 256, device=cuda_device)
    t, keep = model.field_tables()
    _lib.check(lib.nm_selfcheck_field(model.field_handle(), model.mesh_grid.grid.handle, C.byref(t), _lib.ptr(q), _lib.ptr(dirs), P,
                                      _lib.ptr(sdf), _lib.ptr(nab), _lib.ptr(rgb), _lib.ptr(scratch), _lib.ptr(tmp),
                                      _lib.current_stream(cuda_device)), "nm_selfcheck_field", lib)
    with torch.no_grad():
        sdf_m, rgb_m = model.forward(q, dirs)
        _, nab_m = model.forward_with_nablas(q)
    near = np.abs(fx["ds"][:, 0]) < 0.2
    np.testing.assert_allclose(sdf_m[:, 0].cpu().numpy(), sdf.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(rgb_m.cpu().numpy(), rgb.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(nab_m.cpu().numpy()[near], nab.cpu().numpy()[near], atol=1e-5)
    # and the split-half f16 kernels against the fp32 MFMA kernels on the same inputs
    for mode in ("f16x2", "f16x2s"):      # two accumulators / one accumulator
        model.mlp_precision = mode
        with torch.no_grad():
            sdf_h, rgb_h = model.forward(q, dirs)
            _, nab_h = model.forward_with_nablas(q)
        assert bool(torch.isfinite(sdf_h).all() and torch.isfinite(rgb_h).all() and torch.isfinite(nab_h).all())
        np.testing.assert_allclose(sdf_h.cpu().numpy(), sdf_m.cpu().numpy(), atol=2e-6, err_msg=mode)
        np.testing.assert_allclose(rgb_h.cpu().numpy(), rgb_m.cpu().numpy(), atol=2e-6, err_msg=mode)
        np.testing.assert_allclose(nab_h.cpu().numpy()[near], nab_m.cpu().numpy()[near], atol=1e-5, err_msg=mode)
    model.mlp_precision = common.DEFAULT_PRECISION


def test_autograd_path_matches_fused_path(small, cuda_device, torch_mod):
    """Training-style call (grad enabled): HIP K-NN + torch ops; must agree with the fused kernels."""
    torch = torch_mod
    _, _, model = small
    fx = common.golden("field_v3000")
    q, dirs = _t(fx["q"][:512], cuda_device), _t(fx["dirs"][:512], cuda_device)
    sdf_a, rgb_a = model.forward(q.clone(), dirs)
    sdf_b, nab_a = model.forward_with_nablas(q.clone())
    assert sdf_a.requires_grad and nab_a.requires_grad
    with torch.no_grad():
        sdf_f, rgb_f = model.forward(q, dirs)
        _, nab_f = model.forward_with_nablas(q)
    near = np.abs(fx["ds"][:512, 0]) < 0.2
    np.testing.assert_allclose(sdf_a.detach().cpu().numpy(), sdf_f.cpu().numpy(), atol=3e-6)
    np.testing.assert_allclose(rgb_a.detach().cpu().numpy(), rgb_f.cpu().numpy(), atol=3e-6)
    np.testing.assert_allclose(nab_a.detach().cpu().numpy()[near], nab_f.cpu().numpy()[near], atol=1e-5)


# ------------------------------------------------------------------------------ ray set-up
def test_get_rays_kernel_matches_reference(cuda_device, torch_mod):
    from neumesh_amd.rays import get_rays, make_rays
    fx = common.golden("rays_cam")
    H, W = int(fx["H"]), int(fx["W"])
    ro, rd, sel = get_rays(_t(fx["c2w"], cuda_device)[None], _t(fx["intrinsics"], cuda_device)[None], H, W, N_rays=-1)
    assert tuple(rd.shape) == (1, H * W, 3) and np.array_equal(sel[0].cpu().numpy(), np.arange(H * W))
    np.testing.assert_allclose(rd[0].cpu().numpy(), fx["rays_d"], atol=3e-7)
    assert np.array_equal(ro[0].cpu().numpy(), fx["rays_o"])
    o2, d2 = make_rays(fx["c2w"], fx["intrinsics"], H, W, cuda_device, first_pixel=23, count=100)   # a rank's pixel block
    assert torch_mod.equal(d2, rd[0, 23:123]) and torch_mod.equal(o2, ro[0, 23:123])


# ----------------------------------------------------------------------------- renderer
@pytest.mark.parametrize("precision", ["f16x2s", "f16x2", "fp32"])
@pytest.mark.parametrize("tag", ["render_v3000_dtu", "render_v3000_lego"])
def test_render_matches_reference_fixture(small, cuda_device, torch_mod, tag, precision):
    torch = torch_mod
    from neumesh_amd import SingleRenderer
    _, _, model = small
    model.mlp_precision = precision
    rf = common.golden(tag)
    ns = int(rf["N_samples"])
    renderer = SingleRenderer(model)
    kw = dict(batched=True, calc_normal=bool(rf["calc_normal"]), white_bkgd=bool(rf["white_bkgd"]), N_samples=ns, N_importance=ns,
              rayschunk=4096, perturb=False, N_nograd_samples=2048, N_upsample_iters=4, obj_bounding_radius=1.0, bounded_near_far=True,
              H=6, W=12)   # unknown kwargs are swallowed like the reference's **dummy_kwargs
    ro, rd = _t(rf["rays_o"], cuda_device)[None], _t(rf["rays_d"], cuda_device)[None]
    with torch.no_grad():
        rgb, depth, ex = renderer(ro, rd, detailed_output=True, **kw)
        rgb_b, depth_b, ex_b = renderer(ro, rd, detailed_output=False, **{**kw, "rayschunk": 7})
    assert tuple(rgb.shape) == (1, len(rf["rays_o"]), 3) and tuple(depth.shape) == (1, len(rf["rays_o"]))
    assert set(ex_b.keys()) == {"rgb", "depth_volume", "mask_volume"} | ({"normals_volume"} if rf["calc_normal"] else set())
    for k in ("implicit_surface", "radiance", "alpha", "cdf", "visibility_weights", "d_final"):
        assert k in ex
    assert torch.equal(rgb, rgb_b) and torch.equal(depth, depth_b)      # chunk size never changes a pixel
    e = {k: v[0].cpu().numpy() for k, v in ex.items()}
    np.testing.assert_allclose(e["rgb"], rf["rgb"], atol=1e-4)          # BASELINE.json: RGB within 1e-4
    np.testing.assert_allclose(e["depth_volume"], rf["depth_volume"], atol=1e-4)
    np.testing.assert_allclose(e["mask_volume"], rf["mask_volume"], atol=1e-4)
    if rf["calc_normal"]:
        np.testing.assert_allclose(e["normals_volume"], rf["normals_volume"], atol=1e-4)
    np.testing.assert_allclose(e["near_far"], np.concatenate([rf["near"], rf["far"]], 1), atol=2e-6)
    assert compare.psnr(e["rgb"], rf["rgb"]) > 100.0
    worst, unmatched = compare.depth_set_distance(e["d_all"], rf["d_all"])
    assert unmatched < 0.10 and worst < 5e-3                              # oracle/compare.py explains these two
    model.mlp_precision = common.DEFAULT_PRECISION


def test_render_edge_sizes_against_oracle(small, cuda_device, torch_mod):
    """Edge cases of the render call: no ray, one ray, ray counts around the 64-ray workgroup / list-group size, the
    largest sample count the kernels accept (N_samples + N_importance = 256; 257 is refused), near / far bypass.  Every
    case against the CPU oracle on the same rays (RGB within 1e-4) and against a different chunking (bit-equal)."""
    torch = torch_mod
    from neumesh_amd._lib import NeuMeshHipError
    from neumesh_amd.renderer import volume_render
    from oracle import render as orender
    mesh, state, model = small
    orc = common.make_oracle(mesh, state)
    rf = common.golden("render_v3000_dtu")
    ro_all, rd_all = rf["rays_o"], rf["rays_d"]
    with torch.no_grad():
        rgb, depth, ex = volume_render(_t(ro_all[:0], cuda_device), _t(rd_all[:0], cuda_device), model, calc_normal=True, perturb=False, detailed_output=False)
    assert tuple(rgb.shape) == (0, 3) and tuple(depth.shape) == (0,)
    for n in (1, 63, 65):
        ro, rd = ro_all[:n], rd_all[:n]
        want = orender.render_rays(orc, ro, rd, orender.RenderConfig(calc_normal=True))
        with torch.no_grad():
            rgb, depth, ex = volume_render(_t(ro, cuda_device), _t(rd, cuda_device), model, calc_normal=True, perturb=False, detailed_output=False, rayschunk=4096)
            rgb_b, depth_b, _ = volume_render(_t(ro, cuda_device), _t(rd, cuda_device), model, calc_normal=True, perturb=False, detailed_output=False, rayschunk=5)
        assert torch.equal(rgb, rgb_b) and torch.equal(depth, depth_b)
        np.testing.assert_allclose(rgb.cpu().numpy(), want["rgb"], atol=1e-4)
        np.testing.assert_allclose(ex["normals_volume"].cpu().numpy(), want["normals_volume"], atol=1e-4)
    ro, rd = ro_all[:9], rd_all[:9]
    # the largest sample count: 128 + 128 (4 up-sampling iterations of 32)
    want = orender.render_rays(orc, ro, rd, orender.RenderConfig(calc_normal=False, N_samples=128, N_importance=128))
    with torch.no_grad():
        rgb, depth, ex = volume_render(_t(ro, cuda_device), _t(rd, cuda_device), model, calc_normal=False, perturb=False, detailed_output=False,
                                       N_samples=128, N_importance=128)
    np.testing.assert_allclose(rgb.cpu().numpy(), want["rgb"], atol=1e-4)
    with pytest.raises(NeuMeshHipError):
        with torch.no_grad():
            volume_render(_t(ro, cuda_device), _t(rd, cuda_device), model, calc_normal=False, perturb=False, detailed_output=False, N_samples=129, N_importance=128)
    # near / far bypass (renderer.py:172-175)
    want = orender.render_rays(orc, ro, rd, orender.RenderConfig(calc_normal=False, near_bypass=0.6, far_bypass=2.4))
    with torch.no_grad():
        rgb, depth, ex = volume_render(_t(ro, cuda_device), _t(rd, cuda_device), model, calc_normal=False, perturb=False, detailed_output=False,
                                       near_bypass=0.6, far_bypass=2.4)
    np.testing.assert_allclose(rgb.cpu().numpy(), want["rgb"], atol=1e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), want["depth_volume"], atol=1e-4)


def test_wrapper_model_renders_through_staged_path(small, cuda_device, torch_mod):
    """A model that only EXPOSES the five field methods (like the editing tools' wrapper,
    editing/texture_neumesh/texture_neumesh.py:41-51) goes through the staged renderer (per-ray HIP
    stages + the wrapper's methods) and must produce the fused renderer's pixels bit for bit; a wrapper
    that edits the colour (texture swap) must change only the radiance."""
    torch = torch_mod
    from neumesh_amd.renderer import volume_render
    _, _, model = small

    class Wrapper:
        def __init__(self, main, tint=None):
            self.main, self.tint, self.calls = main, tint, []

        def compute_distance(self, xyz):
            self.calls.append(("compute_distance", tuple(xyz.shape)))
            return self.main.compute_distance(xyz)

        def forward_density_only(self, xyz):
            return self.main.forward_density_only(xyz)

        def forward_with_nablas(self, xyz):
            return self.main.forward_with_nablas(xyz)

        def forward_s(self):
            return self.main.forward_s()

        def forward(self, xyz, view_dirs):
            sdf, nabla, ds, idx, w = self.main.forward(xyz, view_dirs, nablas_only=True, return_ds=True)
            feats = self.main.color_features if self.tint is None else self.main.color_features * self.tint
            return sdf, self.main.forward_color(ds, view_dirs, feats, indices=idx, weights=w, nabla=nabla)

    rf = common.golden("render_v3000_dtu")
    ro, rd = _t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device)
    kw = dict(calc_normal=True, perturb=False, N_samples=64, N_importance=64, rayschunk=40, detailed_output=True)
    with torch.no_grad():
        rgb_f, depth_f, ex_f = volume_render(ro, rd, model, **kw)
        wrap = Wrapper(model)
        rgb_w, depth_w, ex_w = volume_render(ro, rd, wrap, **{**kw, "netchunk": 1000})
        rgb_t, _, ex_t = volume_render(ro, rd, Wrapper(model, tint=0.5), **kw)
    assert ("compute_distance", (40, 256, 3)) in wrap.calls            # one probe call per ray chunk (renderer.py:86)
    for k in ("rgb", "depth_volume", "mask_volume", "normals_volume", "d_all", "implicit_surface", "radiance", "implicit_nablas"):
        assert torch.equal(ex_f[k], ex_w[k]), k
    np.testing.assert_allclose(rgb_w.cpu().numpy(), rf["rgb"], atol=1e-4)
    assert torch.equal(ex_t["implicit_surface"], ex_f["implicit_surface"]) and torch.equal(ex_t["d_all"], ex_f["d_all"])
    assert float((rgb_t - rgb_f).abs().max()) > 1e-3                    # the colour edit is visible


def test_render_frame_properties_dtu_scale(dtu_scale, cuda_device, torch_mod):
    """800x800-shape workload on the V=1.4e5 scene: a 64x64 window of the frame. Size-independent
    properties: finite, 0<=acc<=1+eps, chunk invariance, determinism, evaluation-strategy invariance."""
    torch = torch_mod
    from neumesh_amd import synthetic
    from neumesh_amd.renderer import volume_render
    mesh, state, model = dtu_scale
    H = W = 800
    c2w, K = synthetic.orbit_pose(5), synthetic.pinhole_intrinsics(H, W)
    rows = np.arange(368, 432)
    pix = (rows[:, None] * W + np.arange(368, 432)[None, :]).reshape(-1)
    o, d = synthetic.camera_rays(c2w, K, H, W)
    o, d = o[pix], d[pix]
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=False, detailed_output=False)
    with torch.no_grad():
        rgb, depth, ex = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, rayschunk=4096, **kw)
        rgb2, depth2, ex2 = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, rayschunk=1000, **kw)
    assert torch.equal(rgb, rgb2) and torch.equal(depth, depth2) and torch.equal(ex["normals_volume"], ex2["normals_volume"])
    assert bool(torch.isfinite(rgb).all()) and bool(torch.isfinite(depth).all())
    acc = ex["mask_volume"]
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    assert float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0 + 1e-5
    # (parity at this scale is pinned by test_render_headline_scale_matches_reference_fixture below)
    # chained, warm-started tiles of the regular-grid passes (active from 8192 rays per call) must not
    # change a bit relative to independent tiles
    rows = np.arange(336, 464)
    pix = (rows[:, None] * W + np.arange(336, 464)[None, :]).reshape(-1)
    o, d = synthetic.camera_rays(c2w, K, H, W)
    o, d = o[pix], d[pix]
    import os
    old = os.environ.get("NEUMESH_CHAIN_TILES")
    try:
        with torch.no_grad():
            os.environ["NEUMESH_CHAIN_TILES"] = "1"
            a_rgb, a_depth, a_ex = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, rayschunk=16384, **kw)
            os.environ["NEUMESH_CHAIN_TILES"] = "32"
            b_rgb, b_depth, b_ex = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, rayschunk=16384, **kw)
    finally:
        if old is None:
            os.environ.pop("NEUMESH_CHAIN_TILES", None)
        else:
            os.environ["NEUMESH_CHAIN_TILES"] = old
    assert torch.equal(a_rgb, b_rgb) and torch.equal(a_depth, b_depth) and torch.equal(a_ex["normals_volume"], b_ex["normals_volume"])
    # bounded near/far from the first / last hit only (nm_probe_bounds_kernel) == all 256 probes + reduction
    kwd = dict(kw, detailed_output=True)
    try:
        with torch.no_grad():
            os.environ["NEUMESH_FULL_PROBES"] = "1"
            c_rgb, _, c_ex = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, rayschunk=16384, **kwd)
            os.environ.pop("NEUMESH_FULL_PROBES")
            d_rgb, _, d_ex = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, rayschunk=16384, **kwd)
    finally:
        os.environ.pop("NEUMESH_FULL_PROBES", None)
    assert torch.equal(c_ex["near_far"], d_ex["near_far"]) and torch.equal(c_rgb, d_rgb) and torch.equal(c_rgb, b_rgb)
    nf = d_ex["near_far"].cpu().numpy()
    assert (nf[:, 1] > nf[:, 0]).all() and len(np.unique(nf[:, 0])) > 100   # a mix of hit / grazing / missing rays
    # mid-points of weight exactly 0 are not evaluated (default) == every mid-point evaluated; and the
    # depth-bucket lists (16- or 32-ray groups, or none) only decide which lane computes what
    variants = []
    try:
        with torch.no_grad():
            for env in ({"NEUMESH_NO_ZERO_SKIP": "1"}, {"NEUMESH_MID_GROUP": "16"}, {"NEUMESH_MID_GROUP": "32"}, {"NEUMESH_NO_MID_ORDER": "1"},
                        {"NEUMESH_NO_RAY_SORT": "1"},    # (rays processed in the caller's order instead of Morton order)
                        {"NEUMESH_EAGER_NABLAS": "1"}):  # (sample-point nablas inside the sampling passes, all of them)
                os.environ.update(env)
                variants.append(volume_render(_t(o, cuda_device), _t(d, cuda_device), model, rayschunk=16384, **kw))
                for k in env:
                    os.environ.pop(k)
    finally:
        for k in ("NEUMESH_NO_ZERO_SKIP", "NEUMESH_MID_GROUP", "NEUMESH_NO_MID_ORDER", "NEUMESH_NO_RAY_SORT", "NEUMESH_EAGER_NABLAS"):
            os.environ.pop(k, None)
    for v_rgb, v_depth, v_ex in variants:
        assert torch.equal(v_rgb, b_rgb) and torch.equal(v_depth, b_depth) and torch.equal(v_ex["mask_volume"], b_ex["mask_volume"])
        assert torch.equal(v_ex["normals_volume"], b_ex["normals_volume"])
    w = d_ex["visibility_weights"]
    assert 0.2 < float((w == 0).float().mean()) < 0.9   # the skip is exercised: a large share of exact zeros



def test_single_product_f16_mode_is_reduced_precision_and_quantified(surf_scale, cuda_device, torch_mod):
    """mlp_precision 'f16' (one f16 MFMA per product: the "bf16 MLP"-class mode of BASELINE configs[1]) on the surface
    scene's fixture rays: finite, the same coverage classes, PSNR >= 30 dB against the reference -- and NOT within the
    1e-4 RGB bound (s = 400 amplifies an 11-bit SDF), which is why it is never a default.  Prints the error it has."""
    torch = torch_mod
    from neumesh_amd.renderer import volume_render
    mesh, state, model = surf_scale
    f = common.golden("render_v140k_surf")
    ro, rd = _t(f["rays_o"], cuda_device), _t(f["rays_d"], cuda_device)
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=False, rayschunk=65536, detailed_output=False)
    try:
        model.mlp_precision = "f16"
        with torch.no_grad():
            rgb, depth, ex = volume_render(ro, rd, model, **kw)
        assert model.mlp_precision == "f16"      # (no fp16-range fallback happened)
    finally:
        model.mlp_precision = common.DEFAULT_PRECISION
    g = rgb.cpu().numpy().reshape(-1, 3)
    assert np.isfinite(g).all()
    err = np.abs(g - f["rgb"]).max(-1)
    psnr = compare.psnr(g, f["rgb"])
    print(f"single-product f16 MLP vs the reference: max |rgb| error {err.max():.2e}, median {np.median(err):.1e}, "
          f"rays within 1e-4: {100 * (err <= 1e-4).mean():.1f} %, PSNR {psnr:.1f} dB")
    assert psnr >= 30.0 and np.median(err) <= 2e-2
    assert (err > 1e-4).mean() > 0.05, "the single-product mode unexpectedly meets the fp32 bound: report it as such"
    acc = ex["mask_volume"].cpu().numpy().reshape(-1)
    assert ((acc < 1e-3) == (f["mask_volume"] < 1e-3)).mean() >= 0.98


@pytest.mark.parametrize("fixture", ["render_v140k_surf", "render_v140k_dtu"])
def test_per_network_precision_single_product_colour_is_quantified(surf_scale, dtu_scale, cuda_device, torch_mod, fixture):
    """mlp_precision '<default>+f16col' (VERDICT r3 item 3): the geometry network -- whose error the sigmoid at s = 400 amplifies --
    keeps the split-half arithmetic, the colour network -- damped by the output sigmoid's slope <= 1/4 -- takes ONE f16 product.
    Geometry outputs (depth, acc, normals) are bit-identical to the default mode's; the colour error is measured against the
    default-mode frame and against the reference fixtures on both headline-scale scenes (the surface scene has the colour head x 12)
    and bounded: it is a documented, error-quantified mode and NOT the default -- on the surface scene its worst ray uses up the
    1e-4 budget (tools/mlp_error_budget.py predicts 8.6e-5 at the field level)."""
    torch = torch_mod
    from neumesh_amd.renderer import volume_render
    if not common.DEFAULT_PRECISION.startswith("f16x2"):
        pytest.skip("'+f16col' is a variant of the split-half modes (suite default here: " + common.DEFAULT_PRECISION + ")")
    mesh, state, model = surf_scale if fixture.endswith("surf") else dtu_scale
    f = common.golden(fixture)
    ro, rd = _t(f["rays_o"], cuda_device), _t(f["rays_d"], cuda_device)
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=False, rayschunk=65536, detailed_output=False)
    mode = common.DEFAULT_PRECISION.split("+")[0] + "+f16col"
    try:
        with torch.no_grad():
            rgb0, depth0, ex0 = volume_render(ro, rd, model, **kw)
            model.mlp_precision = mode
            rgb, depth, ex = volume_render(ro, rd, model, **kw)
        assert model.mlp_precision == mode
    finally:
        model.mlp_precision = common.DEFAULT_PRECISION
    assert torch.equal(depth, depth0) and torch.equal(ex["mask_volume"], ex0["mask_volume"]) and torch.equal(ex["normals_volume"], ex0["normals_volume"])
    d_mode = (rgb - rgb0).abs().max(-1).values.reshape(-1).cpu().numpy()
    g = rgb.cpu().numpy().reshape(-1, 3)
    err, err0 = np.abs(g - f["rgb"]).max(-1), np.abs(rgb0.cpu().numpy().reshape(-1, 3) - f["rgb"]).max(-1)
    print(f"{mode} on {fixture}: colour change against the default mode max {d_mode.max():.2e}, median {np.median(d_mode):.1e}; vs the reference: "
          f"rays within 1e-4 {100 * (err <= 1e-4).mean():.2f} % (default mode {100 * (err0 <= 1e-4).mean():.2f} %), PSNR {compare.psnr(g, f['rgb']):.1f} dB")
    assert d_mode.max() <= 3e-4 and np.median(d_mode) <= 3e-5
    assert (err <= 1e-4).mean() >= (err0 <= 1e-4).mean() - 0.05


def test_render_headline_scale_matches_reference_fixture(dtu_scale, cuda_device, torch_mod):
    """BASELINE configs[1] scale, pinned to the REFERENCE ITSELF: tests/golden/render_v140k_dtu.npz holds
    1536 strided rays of frame 0 of the 800x800 orbit rendered by the imported reference at V = 140 000
    (oracle/gen_golden.py:gen_scale_fixture), the stages a ray can be traced through and the
    reference's own sensitivity to a 1-ulp nudge of its input directions.  Gates (all can fail):
      * rays whose 128 sample depths are bit-identical to the reference's: |rgb| error <= 1e-4
      * median error <= 1e-6, PSNR >= 60 dB, depth / acc / normals likewise
      * rays beyond 1e-4 (rgb; depth / acc / normals likewise) must be rays the reference itself moves under one of 32 last-bit perturbations
        of the ray directions (tests/golden/render_v140k_dtu_sens.npz, _paired_tail_gate)
      * the field evaluated on the reference's OWN sample points: |sdf| error <= 3e-6
      * production call (detailed_output=False: ray sort + zero-weight skip) == detailed call, bit for bit
    For every ray beyond 1e-4 the first stage whose output leaves the last bit is printed."""
    torch = torch_mod
    from neumesh_amd.renderer import make_render_cfg, render_rays_staged, volume_render
    mesh, state, model = dtu_scale
    f = common.golden("render_v140k_dtu")
    assert int(f["V"]) == mesh.num_vertices
    ro, rd = _t(f["rays_o"], cuda_device), _t(f["rays_d"], cuda_device)
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=False, rayschunk=65536)
    with torch.no_grad():
        rgb, depth, ex = volume_render(ro, rd, model, detailed_output=True, **kw)
        rgb_p, depth_p, ex_p = volume_render(ro, rd, model, detailed_output=False, **kw)
    assert torch.equal(rgb, rgb_p) and torch.equal(depth, depth_p) and torch.equal(ex["mask_volume"], ex_p["mask_volume"])
    assert torch.equal(ex["normals_volume"], ex_p["normals_volume"])
    g = {k: v.cpu().numpy() for k, v in ex.items()}
    err = np.abs(g["rgb"] - f["rgb"]).max(-1)
    self_err = f["self_err_1ulp"]
    same = np.all(g["d_all"] == f["d_all"], axis=1)
    bad = np.nonzero(err > 1e-4)[0]
    print(f"headline-scale parity vs the reference: {len(err)} rays, median {np.median(err):.1e}, max {err.max():.2e}, "
          f"PSNR {compare.psnr(g['rgb'], f['rgb']):.1f} dB, rays > 1e-4: {len(bad)} (reference vs itself + 1 ulp: {int((self_err > 1e-4).sum())}), "
          f"bit-identical sample sets: {int(same.sum())} rays, max error among them {err[same].max() if same.any() else float('nan'):.2e}")
    if len(bad):   # diagnostics: where does a diverging ray leave the reference?
        cfg = make_render_cfg(calc_normal=True)
        tr = {}
        with torch.no_grad():
            render_rays_staged(model, ro[bad], rd[bad], cfg, 65536, 1 << 20, trace=tr)
        got = {"near_far": tr["near_far"][0].cpu().numpy(), "sdf_coarse": tr["sdf_coarse"][0].cpu().numpy()}
        for i, dd in enumerate(tr["d_iter"][0]):
            got[f"d_iter{i + 1}"] = dd.cpu().numpy()
        want = {"near_far": f["near_far"][bad], "sdf_coarse": f["sdf_coarse"][bad], "d_iter1": f["d_iter1"][bad],
                "d_iter2": f["d_iter2"][bad], "d_iter3": f["d_iter3"][bad], "d_iter4": f["d_all"][bad]}
        for j, r in enumerate(bad):
            name, ulps, cnt = compare.first_divergent_stage(got, want, j)
            print(f"  ray {r}: |rgb| error {err[r]:.2e} (reference's own 1-ulp sensitivity {self_err[r]:.2e}); first divergent stage: "
                  f"{name} ({cnt} entries, up to {ulps:.1f} last-place units)")
    assert same.sum() >= 100, "too few rays with bit-identical sample sets for the tight gate to mean anything"
    assert err[same].max() <= 1e-4
    assert np.median(err) <= 1e-6
    assert compare.psnr(g["rgb"], f["rgb"]) >= 60.0
    # the tail, paired ray by ray with the reference's own movement under 32 last-bit perturbations of the rays (round 6: no "+ 1 %" allowance)
    sens = common.golden("render_v140k_dtu_sens")
    assert np.array_equal(sens["self_err"][0], self_err)
    _paired_tail_gate(err, sens, "render_v140k_dtu")
    for key, tol in (("depth_volume", 2e-4), ("mask_volume", 1e-4), ("normals_volume", 1e-4)):
        e = np.abs(g[key] - f[key]).reshape(len(err), -1).max(-1)
        assert np.median(e) <= 1e-6, key
        _paired_tail_gate(e, sens, "render_v140k_dtu", key="self_err_" + key, tol=tol)
    # the field on the reference's own sample points (no sampling differences involved)
    dn = f["rays_d"] / np.linalg.norm(f["rays_d"], axis=-1, keepdims=True)
    pts = (f["rays_o"][:, None, :] + dn[:, None, :] * f["d_all"][..., None]).astype(np.float32)
    with torch.no_grad():
        sdf_g = model.forward_density_only(_t(pts, cuda_device))[..., 0].cpu().numpy()
    # (the reference forms its points as o + d * depth with its own normalised d: 1-ulp position differences
    #  move the sdf by up to ~|nabla| * 1e-7)
    assert np.abs(sdf_g - f["sdf_all"]).max() <= 3e-6
    nf = g["near_far"]
    assert np.abs(nf - f["near_far"]).max() <= 2e-6


@pytest.fixture(scope="module")
def surf_scale(cuda_device):
    mesh = common.scene_mesh(140000)
    state = common.surface_state(mesh)
    return mesh, state, common.make_model(mesh, state, cuda_device)


def _paired_tail_gate(err, sens, label, max_outside=2, key="self_err", tol=1e-4, max_over=2):
    """The end-to-end tail against the reference's OWN spread, ray by ray.  sens[key] [S, n]: the imported reference
    against itself under S independent last-bit perturbations of the ray directions (oracle/gen_golden.py *sens; key "self_err" = rgb,
    "self_err_depth_volume" / "_mask_volume" / "_normals_volume" = the other outputs).  own[r] = the largest move of ray r over the seeds;
    `unstable` = rays the reference itself moves by more than `tol` under SOME seed.
      (a) product rays beyond tol must be reference-unstable rays -- at most `max_outside` exceptions;
      (b) per ray: error <= max(4 x own[r], 10 x tol) -- a ray the reference holds still may not move by more than 10 x tol here, an unstable
          ray by no more than four times what the reference itself does to it -- at most `max_over` exceptions;
      (c) no more rays beyond tol than the reference's worst seed plus a quarter (at least 2): a 33rd draw from the same distribution exceeds the
          maximum of 32 with probability 1/33, and the gate runs on four outputs in three arithmetics.
    The allowances are what the reference's OWN seeds need when each is held against the other 31 (leave-one-out over the four *_sens fixtures and
    four outputs: at most 2 rays outside the others' unstable set, at most 1 ray over its own limit -- a chaotic ray's movement is heavy-tailed);
    the leave-one-out figures of this fixture are printed beside the product's.  Both allowances are 2: a change of the MLP ARITHMETIC (the fp32
    kernels against torch's fp32: sdf values 5e-7 apart at every sample) is not the same perturbation as a last-bit nudge of the ray directions,
    and moves one or two rays the nudges leave alone (measured: fp32 mode, rays 1151 and 1167 of the surface fixture).
    Returns (unstable, stable_under_all_seeds)."""
    own = sens[key].max(0)
    unstable = own > tol
    bad = err > tol
    outside = np.nonzero(bad & ~unstable)[0]
    lim = np.maximum(4.0 * own, 10.0 * tol)
    over = np.nonzero(err > lim)[0]
    counts = (sens[key] > tol).sum(1)
    loo_out, loo_over = 0, 0                                   # the reference against itself, each seed vs the others
    for i in range(sens[key].shape[0]):
        others = np.delete(sens[key], i, 0).max(0)
        loo_out = max(loo_out, int(((sens[key][i] > tol) & ~(others > tol)).sum()))
        loo_over = max(loo_over, int((sens[key][i] > np.maximum(4.0 * others, 10.0 * tol)).sum()))
    print(f"  [{label}] paired tail gate ({key} > {tol:g}) over {sens[key].shape[0]} reference seeds: reference-unstable rays {int(unstable.sum())} (per seed {int(counts.min())}..{int(counts.max())}); "
          f"product rays beyond: {int(bad.sum())}, of them outside the unstable set: {len(outside)} {[(int(r), float(f'{err[r]:.1e}'), float(f'{own[r]:.1e}')) for r in outside[:8]]}; "
          f"rays over their own limit max(4 x own, 10 x tol): {len(over)} {[(int(r), float(f'{err[r]:.1e}'), float(f'{own[r]:.1e}')) for r in over[:8]]}; max {err.max():.2e} "
          f"(reference seeds leave-one-out: outside <= {loo_out}, over <= {loo_over})")
    if os.environ.get("NEUMESH_PARITY_DUMP"):
        os.makedirs(os.environ["NEUMESH_PARITY_DUMP"], exist_ok=True)
        np.save(os.path.join(os.environ["NEUMESH_PARITY_DUMP"], f"parity_tail_{label}.{key}.npy"), err)
    assert len(outside) <= max_outside, (label, key, [(int(r), float(err[r]), float(own[r])) for r in outside])
    assert len(over) <= max_over, (label, key, [(int(r), float(err[r]), float(own[r])) for r in over])
    assert int(bad.sum()) <= int(counts.max()) + max(2, int(counts.max()) // 4), (label, key, int(bad.sum()), int(counts.max()))
    return unstable, own <= 0.01 * tol


def _scene_against_reference_fixture(model, digest, fixture, precision, cuda_device, torch, classes=True):
    """The gates of the headline-scale end-to-end tests (see test_render_surface_scene_matches_reference_fixture); `digest`: sha256 of the
    weights the caller built `model` from -- it must be the one the fixture was rendered with."""
    from neumesh_amd.renderer import make_render_cfg, render_at_depths, volume_render
    f = common.golden(fixture)
    sens = common.golden(fixture + "_sens")
    ns, ni, white = (int(f["N_samples"]), int(f["N_importance"]), bool(f["white_bkgd"])) if "N_samples" in f.files else (64, 64, False)
    assert str(f["state_sha256"]) == digest and str(sens["state_sha256"]) == digest
    assert abs(float(model.forward_s()) - float(f["s"])) <= 1e-3 * max(1.0, float(f["s"]) / 400.0)
    label = f"{fixture}.{precision}"
    ro, rd = _t(f["rays_o"], cuda_device), _t(f["rays_d"], cuda_device)
    n = len(f["rgb"])
    old_precision = model.mlp_precision
    model.mlp_precision = precision
    try:
        # (1) behind the sampler
        with torch.no_grad():
            tail = render_at_depths(model, ro, rd, _t(f["d_all"], cuda_device),
                                    make_render_cfg(calc_normal=True, N_samples=ns, N_importance=ni, white_bkgd=white), detailed=True)
        t = {k: v.cpu().numpy() for k, v in tail.items()}
        worst = {}
        for key, tol in (("rgb", 1e-4), ("mask_volume", 1e-4), ("normals_volume", 1e-4), ("depth_volume", 2e-4)):
            e = np.abs(t[key] - f[key]).reshape(n, -1).max(-1)
            worst[key] = float(e.max())
            assert e.max() <= tol, (key, float(e.max()), int(e.argmax()))
        e_sdf = np.abs(t["implicit_surface"] - f["sdf_all"])
        off = np.argwhere(e_sdf > 3e-6)
        print(f"{label}, on the reference's own depths ({n} rays): max errors {worst}, |sdf| {e_sdf.max():.2e}, points beyond 3e-6: {len(off)} of {e_sdf.size}")
        # (4) the field on the reference's own sample points: 3e-6 on every point -- except where the reference's field is itself discontinuous or
        # ill-conditioned in the last bit of the POSITION (the product forms o + d * depth with its own normalised d: 1-ulp differences): a point
        # whose 8th and 9th nearest vertices are (nearly) equidistant takes another neighbour set (seen on the trained field: ONE of 196 608
        # points, an exact fp32 tie, 2e-4), and a point almost on a vertex has weights 1 / (d + 1e-7) that move with the last bit (1e-5).
        # At most 8 such points, each one verified to be of that kind with the product's own exact K-NN, none beyond 2e-3.
        assert len(off) <= 8, len(off)
        if len(off):
            from neumesh_amd.mesh_grid import knn as hip_knn
            dn_ = f["rays_d"] / np.linalg.norm(f["rays_d"], axis=-1, keepdims=True)
            pts_off = np.stack([(f["rays_o"][r] + dn_[r] * f["d_all"][r, j]).astype(np.float32) for r, j in off])
            _, d2o = hip_knn(model.mesh_grid.grid, _t(pts_off, cuda_device), 9)
            d2o = d2o.cpu().numpy()
            for (r, j), row in zip(off, d2o):
                tie = (row[8] - row[7]) <= 1e-5 * row[7]                 # on a boundary between two neighbour sets
                near_vertex = np.sqrt(row[0]) <= 0.2 * np.sqrt(row[7])   # dominated by one vertex: 1 / (d + 1e-7) weights are steep
                print(f"    point (ray {r}, sample {j}): |sdf| error {e_sdf[r, j]:.2e}, d2[0] {row[0]:.3e}, d2[7..8] {row[7]:.6e} {row[8]:.6e} -> {'tie' if tie else 'near a vertex' if near_vertex else 'UNEXPLAINED'}")
                assert e_sdf[r, j] <= (2e-3 if tie else 2e-5), (int(r), int(j), float(e_sdf[r, j]))
                assert tie or near_vertex, (int(r), int(j), row.tolist())
        # (2), (3) end to end
        kw = dict(calc_normal=True, N_samples=ns, N_importance=ni, white_bkgd=white, perturb=False, rayschunk=65536)
        with torch.no_grad():
            rgb, depth, ex = volume_render(ro, rd, model, detailed_output=True, **kw)
            rgb_p, depth_p, ex_p = volume_render(ro, rd, model, detailed_output=False, **kw)
        assert model.mlp_precision == precision, "the call left the fp16 range and fell back to the fp32 kernels"
        assert torch.equal(rgb, rgb_p) and torch.equal(depth, depth_p) and torch.equal(ex["mask_volume"], ex_p["mask_volume"])
        assert torch.equal(ex["normals_volume"], ex_p["normals_volume"])
        g = {k: v.cpu().numpy() for k, v in ex.items()}
        err = np.abs(g["rgb"] - f["rgb"]).max(-1)
        self_err = f["self_err_1ulp"]
        assert np.array_equal(sens["self_err"][0], self_err)
        acc_r, acc_g = f["mask_volume"], g["mask_volume"]
        print(f"{label} end to end: median {np.median(err):.1e}, max {err.max():.2e}, PSNR {compare.psnr(g['rgb'], f['rgb']):.1f} dB, "
              f"rays > 1e-4: {int((err > 1e-4).sum())}; acc == 0: {int((acc_g == 0).sum())}, partial: {int(((acc_g >= 1e-3) & (acc_g <= 0.999)).sum())}, "
              f"opaque: {int((acc_g > 0.999).sum())}")
        bad = np.nonzero(err > 1e-4)[0]
        if len(bad) and "d_iter1" in f.files:   # attribution: the first stage at which a diverging ray leaves the reference's last bit
            from neumesh_amd.renderer import render_rays_staged
            tr = {}
            with torch.no_grad():
                render_rays_staged(model, ro[bad], rd[bad], make_render_cfg(calc_normal=True, N_samples=ns, N_importance=ni, white_bkgd=white), 65536, 1 << 20, trace=tr)
            got = {"near_far": tr["near_far"][0].cpu().numpy(), "sdf_coarse": tr["sdf_coarse"][0].cpu().numpy()}
            for i, dd in enumerate(tr["d_iter"][0]):
                got[f"d_iter{i + 1}"] = dd.cpu().numpy()
            want = {"near_far": f["near_far"][bad], "sdf_coarse": f["sdf_coarse"][bad], "d_iter1": f["d_iter1"][bad],
                    "d_iter2": f["d_iter2"][bad], "d_iter3": f["d_iter3"][bad], "d_iter4": f["d_all"][bad]}
            stages = {}
            for j, r in enumerate(bad):
                name, ulps, cnt = compare.first_divergent_stage(got, want, j)
                stages[name] = stages.get(name, 0) + 1
            print(f"  first divergent stage of the {len(bad)} rays beyond 1e-4: {stages}")
            # (a ray listed under None has every sampling stage within the last bit / the field's 3e-6 of the reference's: its error is the
            #  field tolerance times s at a crossing, bounded by gate (1) above; it must stay the exception)
            assert stages.get(None, 0) <= max(2, len(bad) // 4), stages
        if classes:
            assert int((acc_r == 0).sum()) >= 0.2 * n and int(((acc_r >= 1e-3) & (acc_r <= 0.999)).sum()) >= 0.08 * n and int((acc_r > 0.999).sum()) >= 0.5 * n
        assert float(f["rgb"].std()) > 0.1
        assert np.array_equal(acc_g == 0, acc_r == 0)
        assert np.abs(g["near_far"] - f["near_far"]).max() <= 2e-6
        assert np.median(err) <= 1e-6
        unstable, still = _paired_tail_gate(err, sens, label)
        for key, tol in (("depth_volume", 2e-4), ("mask_volume", 1e-4), ("normals_volume", 1e-4)):
            e = np.abs(g[key] - f[key]).reshape(n, -1).max(-1)
            assert np.median(e) <= 1e-6, key
            _paired_tail_gate(e, sens, label, key="self_err_" + key, tol=tol)       # (paired with the reference's own movement of THIS output)
        # Rays the reference holds still under EVERY seed: with sample depths IDENTICAL to the reference's the bound is north_star's 1e-4 on
        # every such ray.  Depths that differ in the last bits are not enough for that: the reference's field is discontinuous where a
        # point's 8-neighbour set changes (its own secant roots sit on jumps of up to 1.5e-3, tests/golden/surface_v140k_surf.npz), so a sample
        # next to such a boundary can change sides under a 1-ulp move and s turns the jump into a visible alpha change.
        dd = np.abs(g["d_all"] - f["d_all"]).max(-1)
        calm = still & (dd <= 2e-6)
        same = still & (dd == 0)
        print(f"  calm rays (still under all {sens['self_err'].shape[0]} seeds, depths within 2e-6): {int(calm.sum())} (bit-identical depths: {int(same.sum())}), max error among them "
              f"{err[calm].max():.2e} / {err[same].max() if same.any() else 0.0:.2e}; calm rays beyond 1e-4: {int((err[calm] > 1e-4).sum())}")
        assert same.sum() >= 20 and err[same].max() <= 1e-4, (int(same.sum()), float(err[same].max()) if same.any() else None)
        assert calm.sum() >= 0.15 * n and (err[calm] > 1e-4).mean() <= 0.01, (int(calm.sum()), int((err[calm] > 1e-4).sum()))
        assert err[calm].max() <= 1e-3, float(err[calm].max())    # (an absolute cap on the calm rays' outliers, in every arithmetic)
        return {"err": err, "g": g, "f": f}
    finally:
        model.mlp_precision = old_precision          # (a failing gate must not leave the shared model in another arithmetic)


@pytest.mark.parametrize("fixture,precision", [("render_v140k_surf", "f16x2s"), ("render_v140k_surf", "f16x2"), ("render_v140k_surf", "fp32"),
                                               ("render_v140k_surf_c3", "f16x2s")])
def test_render_surface_scene_matches_reference_fixture(surf_scale, cuda_device, torch_mod, fixture, precision):
    """(render_v140k_surf_c3: the same scene in BASELINE configs[3]'s shape -- 32 + 32 samples, white background -- 1024 rays, so that
    config 3 is pinned to the reference at headline scale too; VERDICT r2 weak #1.)
    The headline shape on a scene WITH A SURFACE: tests/golden/render_v140k_surf.npz = 1536 strided
    rays of frame 0 rendered by the imported reference on the weights of synthetic.surface_mlp_state (sdf = ds + bump,
    s = 400): 356 rays with acc == 0 (they miss: near/far fall back to the bounding sphere), 130 partially covered, 911
    opaque, rgb std 0.25.  A sharp crossing makes the reference's sample placement sensitive to the last bit (its own
    render with directions nudged by 1 ulp moves 18 rays by > 1e-4, one by 1.6e-3), so the gates are
      (1) everything behind the sampler, on the REFERENCE'S OWN 128 depths per ray (render_at_depths): rgb / acc /
          normals <= 1e-4 and depth <= 2e-4 on EVERY ray;
      (2) end to end: median <= 1e-6, near/far <= 2e-6, coverage classes (acc == 0 / partial / opaque) of every ray equal to the
          reference's, and the tail PAIRED with the reference's own spread (round 6): tests/golden/<fixture>_sens.npz
          holds the imported reference against itself under 32 independent last-bit perturbations of the ray directions; every MLP
          arithmetic of the product (f16x2s, f16x2, fp32: one test each) must keep its rays beyond 1e-4 INSIDE the set of rays the
          reference itself moves (_paired_tail_gate: <= 2 exceptions, per-ray limit 4 x the ray's own movement or 1e-3, count <= worst seed);
      (3) production call (ray sort, first/last-hit probe walk, zero-weight skip) == detailed call, bit for bit;
      (4) the field on the reference's own sample points: |sdf| <= 3e-6."""
    mesh, state, model = surf_scale
    assert int(common.golden(fixture)["V"]) == mesh.num_vertices
    digest = common.state_digest({k: v for k, v in state.items() if k not in ("geometry_features", "color_features", "indicator_vector")})
    _scene_against_reference_fixture(model, digest, fixture, precision, cuda_device, torch_mod)


@pytest.fixture(scope="module")
def trained_scale(cuda_device):
    mesh = common.scene_mesh(140000)
    state = common.trained_state()
    return mesh, state, common.make_model(mesh, state, cuda_device)


@pytest.mark.parametrize("precision", ["f16x2s", "f16x2", "fp32"])
def test_render_trained_field_matches_reference_fixture(trained_scale, cuda_device, torch_mod, precision):
    """A TRAINED field (VERDICT r5 missing #1 / item 1): tests/golden/trained_v140k.pt = 20 000 iterations of the reference's training
    recipe on an analytic scene at V = 140 000 (tools/train_field.py: Adam 5e-4 warm-up + cosine, full loss set incl. distillation from the
    analytic teacher, s = 1000 frozen as the teacher's; held-out PSNR 33 dB), in utils/checkpoints.py's layout.  The fixture
    render_v140k_trained.npz is the IMPORTED REFERENCE on that file loaded as render.py:287-288 does (oracle/gen_golden.py `trained`), with
    its 32-seed self-sensitivity.  Same gates as the hand-built surface scene, in all three precise arithmetics; in addition the call must
    not trip the fp16-range flag (the split-half default would silently become the fp32 kernels), and the margins of the default
    arithmetic on these weights are printed (largest activation per layer, tangent operands, sin/cos arguments)."""
    torch = torch_mod
    mesh, state, model = trained_scale
    digest = common.state_digest(state)
    out = _scene_against_reference_fixture(model, digest, "render_v140k_trained", precision, cuda_device, torch, classes=False)
    assert abs(float(model.forward_s()) - 1000.0) <= 1.0
    if precision == "f16x2s":
        m = common.field_margins(model, _t(out["f"]["rays_o"], cuda_device), _t(out["f"]["rays_d"], cuda_device), _t(out["f"]["d_all"], cuda_device))
        print("  margins of the split-half default on the trained field:", {k: (float(f"{v:.3g}") if isinstance(v, float) else v) for k, v in m.items()})
        assert m["max_abs_operand_value_rows"] < 0.5 * 65504 and m["max_abs_operand_tangent_rows"] < 0.5 * 65504
        assert m["sincos_fast_range_exceeded_share"] <= 0.01


@pytest.mark.gpu
def test_weight_eps_drops_only_negligible_terms(dtu_scale, cuda_device, torch_mod):
    """nm_render_cfg.weight_eps (the one setting that is not bit-identical): mid-points whose visibility weight is below it
    are not evaluated.  On the headline fixture rays with weight_eps = 1e-10: depth and acc bit-identical to the exact
    render (their weights come from the sample SDFs), rgb / normals within (N-1) * weight_eps + fp32 rounding of the sums,
    parity with the reference unchanged, and far fewer mid-points evaluated."""
    torch = torch_mod
    from neumesh_amd import _lib
    from neumesh_amd.renderer import make_render_cfg, render_rays_fused
    mesh, state, model = dtu_scale
    f = common.golden("render_v140k_dtu")
    ro, rd = _t(f["rays_o"], cuda_device), _t(f["rays_d"], cuda_device)
    lib = _lib.load()
    out, pts = {}, {}
    for name, eps in (("exact", 0.0), ("eps", 1e-10)):
        cfg = make_render_cfg(calc_normal=True, weight_eps=eps, flags=0)
        lib.nm_profile_enable(1)
        with torch.no_grad():
            out[name] = render_rays_fused(model, ro, rd, cfg, 65536)
        torch.cuda.synchronize()
        ms, n, u = C.c_double(), C.c_int64(), C.c_int64()
        lib.nm_profile_read(3, C.byref(ms), C.byref(n), C.byref(u))   # kind 3: colour MLP = evaluated mid-points
        pts[name] = u.value
        lib.nm_profile_enable(0)
    a, b = out["exact"], out["eps"]
    assert torch.equal(a["depth_volume"], b["depth_volume"]) and torch.equal(a["mask_volume"], b["mask_volume"])
    d_rgb = float((a["rgb"] - b["rgb"]).abs().max()), float((a["normals_volume"] - b["normals_volume"]).abs().max())
    print(f"weight_eps 1e-10: evaluated mid-points {pts['exact']} -> {pts['eps']}, max |rgb| change {d_rgb[0]:.2e}, normals {d_rgb[1]:.2e}")
    assert d_rgb[0] <= 127e-10 + 2.5e-7 and d_rgb[1] <= 127e-10 + 5e-7   # dropped terms + a few last-place units of the sums
    assert np.abs(b["rgb"].cpu().numpy() - f["rgb"]).max() <= 1e-4
    assert 0 < pts["eps"] < 0.8 * pts["exact"]


@pytest.mark.gpu
def test_frame_assembly_matches_render_py_arithmetic(cuda_device, torch_mod):
    """nm_assemble_frame vs the host arithmetic of render.py:183-184, 219-249 (numpy float32, truncating uint8 cast):
    bit-exact on values inside the cast's defined range, clamped outside; BGR swap; optional outputs."""
    torch = torch_mod
    from neumesh_amd.frames import assemble_images
    rng = np.random.default_rng(5)
    H, W = 37, 53
    rgb = rng.random((H * W, 3), dtype=np.float32)
    rgb[:8] = [[0.0, 1.0, 0.999999], [1.0000001, 0.5, 0.25], [1e-9, 0.0039215689, 0.0039215684], [0.5, 0.5, 0.5],
               [1.5, -0.25, 2.0], [0.99609375, 0.99609381, 0.996], [0.2, 0.4, 0.6], [1.0, 1.0, 1.0]]
    depth = (rng.random(H * W, dtype=np.float32) * 3.0).astype(np.float32)
    nrm = rng.standard_normal((H * W, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    out = assemble_images(_t(rgb, cuda_device), _t(depth, cuda_device), _t(nrm, cuda_device), H, W)
    out_bgr = assemble_images(_t(rgb, cuda_device), H=H, W=W, bgr=True)

    def integerify(img):   # render.py:183-184, with the clamp where numpy's cast is undefined
        return np.clip(np.trunc(img * np.float32(255.0)), 0, 255).astype(np.uint8)
    want_rgb = integerify(rgb).reshape(H, W, 3)
    want_depth = integerify(depth / depth.max()).reshape(H, W, 1)
    want_nrm = integerify(nrm / np.float32(2.0) + np.float32(0.5)).reshape(H, W, 3)
    inside = ((rgb * np.float32(255.0) >= 0) & (rgb * np.float32(255.0) < 256)).all()
    assert not inside   # the clamped cases are part of the test
    assert np.array_equal(out["rgb"].cpu().numpy(), want_rgb)
    assert np.array_equal(out["depth"].cpu().numpy(), want_depth)
    assert np.array_equal(out["normal"].cpu().numpy(), want_nrm)
    assert set(out_bgr) == {"rgb"} and np.array_equal(out_bgr["rgb"].cpu().numpy(), want_rgb[..., ::-1])
    ok = (rgb * np.float32(255.0) < 256).all(-1) & (rgb >= 0).all(-1)
    assert np.array_equal(out["rgb"].cpu().numpy().reshape(-1, 3)[ok], (rgb[ok] * 255.0).astype(np.uint8))   # numpy's own cast


@pytest.mark.gpu
def test_config5_stress_kernels_1M_vertices_256d(cuda_device, torch_mod):
    """BASELINE config 5: V = 1 000 000 vertices, one 256-d feature table, kernels = K-NN +
    gather-interpolate only (nm_distance_interpolate).  Coherent queries (points of adjacent camera
    rays near the surface) and scattered ones, against the oracle: indices bit-exact, weights / ds /
    interpolated features within fp32 rounding of the declared arithmetic; linearity of the
    interpolation in the table as a size-independent property on 2^20 queries."""
    torch = torch_mod
    from neumesh_amd import synthetic
    from neumesh_amd.mesh_grid import MeshGrid
    from oracle import field as ofield
    V, dim = 1_000_000, 256
    mesh = synthetic.fibonacci_blob(V)
    grid = MeshGrid(mesh, cuda_device)
    gen = torch.Generator(device=cuda_device)
    gen.manual_seed(5)
    table = torch.randn((V, dim), generator=gen, device=cuda_device)
    # coherent: 64x64 pixel window, one point per ray where the ray meets the r = 0.75 sphere (+ jitter)
    H = W = 4096
    c2w, K = synthetic.orbit_pose(2), synthetic.pinhole_intrinsics(H, W)
    def window(r0, r1, c0, c1):  # rays of the pixel window [r0,r1) x [c0,c1), row-major, unit directions
        o, d = synthetic.camera_rays(c2w, K, H, W, start=r0 * W, count=(r1 - r0) * W)
        sel = (np.arange(r1 - r0)[:, None] * W + np.arange(c0, c1)[None, :]).reshape(-1)
        return o[sel], orender.normalize(d[sel])

    o, d = window(2000, 2064, 2000, 2064)
    b = (o * d).sum(-1)
    t_hit = -b - np.sqrt(np.maximum(b * b - ((o * o).sum(-1) - 0.75 ** 2), 0.0))
    rng = np.random.default_rng(9)
    q_coh = (o + (t_hit + rng.uniform(-0.02, 0.02, len(o)))[:, None] * d).astype(np.float32)
    q_sct = (mesh.vertices[rng.integers(0, V, 2048)] + 0.05 * rng.standard_normal((2048, 3))).astype(np.float32)
    q = np.concatenate([q_coh, q_sct])
    with torch.no_grad():
        ds, idx, w, feat = grid.compute_distance_interpolate(_t(q, cuda_device), table)
    ridx, rd2 = oknn.knn_kdtree(q, mesh.vertices, 8)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    dis = np.sqrt(rd2)
    rw = (1.0 / (dis + np.float32(1e-7))).astype(np.float32)
    rw = (rw / rw.sum(-1, keepdims=True)).astype(np.float32)
    np.testing.assert_allclose(w.cpu().numpy(), rw, rtol=2e-6, atol=1e-7)
    rows_needed = table[torch.from_numpy(ridx).to(cuda_device)].cpu().numpy().astype(np.float64)   # [Q,8,dim]
    rfeat = (rows_needed * rw.astype(np.float64)[..., None]).sum(-2)
    np.testing.assert_allclose(feat.cpu().numpy(), rfeat, atol=4e-6)
    # size-independent property on 2^20 coherent queries: interpolation is linear in the table
    # (same neighbours, same weights): f(2*T) == 2*f(T) bit for bit (scaling by 2 is exact)
    o2, d2 = window(1536, 2560, 1536, 2560)
    b2 = (o2 * d2).sum(-1)
    t2 = -b2 - np.sqrt(np.maximum(b2 * b2 - ((o2 * o2).sum(-1) - 0.75 ** 2), 0.0))
    q2 = _t((o2 + t2[:, None] * d2).astype(np.float32), cuda_device)
    with torch.no_grad():
        _, i1, w1_, f1 = grid.compute_distance_interpolate(q2, table)
        _, i2, w2_, f2 = grid.compute_distance_interpolate(q2, table * 2.0)
    assert torch.equal(i1, i2) and torch.equal(w1_, w2_) and torch.equal(f1 * 2.0, f2)
    assert bool(torch.isfinite(f1).all()) and bool((w1_.sum(-1) - 1.0).abs().max() < 1e-5)


@pytest.mark.gpu
def test_deformed_mesh_grid_swap_like_deform_model(cuda_device, torch_mod):
    """editing/render_geometry_editing.py:37-67 (deform_model): a new MeshGrid is built on the deformed
    mesh, assigned to model.mesh_grid, and the indicator vectors are replaced by a new nn.Parameter.
    The render must then equal a model built on the deformed mesh from scratch, and match the oracle."""
    torch = torch_mod
    from neumesh_amd import MeshGrid, synthetic
    from neumesh_amd.renderer import volume_render
    mesh = common.scene_mesh(3000)
    state = common.scene_state(mesh)
    model = common.make_model(mesh, state, cuda_device)
    # deformation: anisotropic stretch + shear of the vertices, normals transformed by the inverse transpose
    A = np.array([[1.15, 0.10, 0.0], [0.0, 0.9, 0.05], [0.0, 0.0, 1.05]], np.float32)
    dv = (mesh.vertices @ A.T).astype(np.float32)
    dn = mesh.vertex_normals @ np.linalg.inv(A)
    dn = (dn / np.linalg.norm(dn, axis=-1, keepdims=True)).astype(np.float32)
    dmesh = synthetic.SyntheticMesh(dv, dn)
    new_ind = synthetic.noisy_indicator(dn, 11)
    o, d = synthetic.camera_rays(synthetic.orbit_pose(1), synthetic.pinhole_intrinsics(64, 64), 64, 64)
    sel = np.arange(0, 64 * 64, 13)[:256]
    o, d = o[sel], d[sel]
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=False, detailed_output=False, rayschunk=4096)
    with torch.no_grad():
        before = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, **kw)[0].clone()
        model.mesh_grid = MeshGrid(common.MeshObj(dmesh), cuda_device, distance_method=model.mesh_grid.distance_method)
        model.indicator_vector = torch.nn.Parameter(_t(new_ind, cuda_device))
        rgb, depth, ex = volume_render(_t(o, cuda_device), _t(d, cuda_device), model, **kw)
    dstate = dict(state)
    dstate["indicator_vector"] = new_ind
    fresh = common.make_model(dmesh, dstate, cuda_device)
    with torch.no_grad():
        rgb2, depth2, ex2 = volume_render(_t(o, cuda_device), _t(d, cuda_device), fresh, **kw)
    assert torch.equal(rgb, rgb2) and torch.equal(depth, depth2) and not torch.equal(rgb, before)
    out = orender.render_rays(common.make_oracle(dmesh, dstate), o, d, orender.RenderConfig(calc_normal=True))
    np.testing.assert_allclose(rgb.cpu().numpy(), out["rgb"], atol=1e-4)
    np.testing.assert_allclose(ex["normals_volume"].cpu().numpy(), out["normals_volume"], atol=2e-4)


@pytest.mark.gpu
def test_training_render_forward_and_gradients_match_reference(small, cuda_device, torch_mod):
    """SURVEY 8f rank 3 (trainer.py:75-81): volume_render with autograd enabled -- sample placement on
    the HIP stage kernels without gradients, field + compositing differentiable.  Forward equals the
    fused inference renderer; d loss / d parameter equals the REFERENCE's own backward pass
    (tests/golden/grad_v3000_dtu.npz, produced by oracle/gen_golden.py::gen_grad_fixture)."""
    torch = torch_mod
    from neumesh_amd.renderer import volume_render
    mesh, state, _ = small
    model = common.make_model(mesh, state, cuda_device)   # private copy: gradients are accumulated on it
    model.train()
    rf, gf = common.golden("render_v3000_dtu"), common.golden("grad_v3000_dtu")
    o, d = _t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device)
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=False, detailed_output=False, rayschunk=4096)
    with torch.no_grad():
        rgb0, depth0, ex0 = volume_render(o, d, model, **kw)
    rgb, depth, ex = volume_render(o, d, model, **kw)          # autograd on
    assert rgb.requires_grad
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), rgb0.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(depth.detach().cpu().numpy(), depth0.cpu().numpy(), atol=5e-6)
    np.testing.assert_allclose(ex["normals_volume"].detach().cpu().numpy(), ex0["normals_volume"].cpu().numpy(), atol=5e-5)  # autograd nablas vs closed form
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), gf["rgb"], atol=1e-4)
    loss = (rgb * _t(gf["w_rgb"], cuda_device)).sum() + 0.1 * depth.sum() + 0.05 * ex["mask_volume"].sum() \
        + 0.02 * (ex["normals_volume"] * _t(gf["w_n"], cuda_device)).sum()
    assert abs(float(loss) - float(gf["loss"])) < 2e-3
    loss.backward()
    checked = 0
    for name, p in model.named_parameters():
        key = "grad." + name
        if key not in gf.files:
            continue
        assert p.grad is not None, name
        g = p.grad.detach().cpu().numpy()
        ref_norm = float(gf["norm." + name])
        assert abs(float(np.linalg.norm(g.astype(np.float64))) - ref_norm) <= 2e-3 * ref_norm + 1e-6, name
        if "rows." + name in gf.files:
            g = g[gf["rows." + name]]
        err = np.abs(g - gf[key]).max()
        assert err <= 2e-3 * np.abs(gf[key]).max() + 1e-6, (name, err)
        checked += 1
    assert checked >= 20
    # perturb=True (sample_pdf(det=False)): stochastic sample placement, same estimator
    torch.manual_seed(3)
    with torch.no_grad():
        a = volume_render(o, d, model, **dict(kw, perturb=True))[0]
        torch.manual_seed(3)
        b = volume_render(o, d, model, **dict(kw, perturb=True))[0]
        c = volume_render(o, d, model, **dict(kw, perturb=True))[0]
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert float((a - rgb0).abs().mean()) < 0.05 and bool(torch.isfinite(a).all())


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(N_samples=32, N_importance=32, N_upsample_iters=4, calc_normal=False, white_bkgd=True, bounded_near_far=True),
    dict(N_samples=48, N_importance=24, N_upsample_iters=3, calc_normal=True, white_bkgd=False, bounded_near_far=False),
    dict(N_samples=64, N_importance=0, N_upsample_iters=4, calc_normal=True, white_bkgd=False, bounded_near_far=True),
    dict(N_samples=96, N_importance=96, N_upsample_iters=2, calc_normal=True, white_bkgd=True, bounded_near_far=True),
])
def test_fused_renderer_equals_staged_renderer_other_configs(dtu_scale, cuda_device, torch_mod, cfg):
    """Ragged ray counts and sampling configurations other than the headline one (lego-style 32+32 /
    white background, no importance samples, 3 or 2 up-sampling iterations, sphere near/far, 192 samples per
    ray): the fused renderer -- spatial ray order, first/last-hit probes, depth-bucket lists, zero-weight
    skip -- against the staged renderer, which runs every stage plainly through the model's methods."""
    torch = torch_mod
    from neumesh_amd import synthetic
    from neumesh_amd.renderer import volume_render
    mesh, state, model = dtu_scale

    class Plain:   # exposes only the field methods => volume_render takes the staged path
        def __init__(self, m):
            self.m = m

        def compute_distance(self, xyz):
            return self.m.compute_distance(xyz)

        def forward_density_only(self, xyz):
            return self.m.forward_density_only(xyz)

        def forward_with_nablas(self, xyz):
            return self.m.forward_with_nablas(xyz)

        def forward_s(self):
            return self.m.forward_s()

        def forward(self, xyz, view_dirs):
            return self.m.forward(xyz, view_dirs)

    H = W = 800
    o, d = synthetic.camera_rays(synthetic.orbit_pose(9), synthetic.pinhole_intrinsics(H, W), H, W, start=390 * W, count=20 * W)
    sel = np.arange(0, 20 * W, 7)[:1999]   # 1999 rays: not a multiple of 16 / 64, grazing and missing rays included
    o, d = _t(o[sel], cuda_device), _t(d[sel], cuda_device)
    kw = dict(cfg, perturb=False, detailed_output=False)
    with torch.no_grad():
        rgb_f, depth_f, ex_f = volume_render(o, d, model, rayschunk=1999, **kw)
        rgb_c, depth_c, ex_c = volume_render(o, d, model, rayschunk=777, **kw)     # three chunks on two streams
        rgb_s, depth_s, ex_s = volume_render(o, d, Plain(model), rayschunk=1000, **kw)
    for a, b in ((rgb_f, rgb_c), (depth_f, depth_c), (ex_f["mask_volume"], ex_c["mask_volume"]),
                 (rgb_f, rgb_s), (depth_f, depth_s), (ex_f["mask_volume"], ex_s["mask_volume"])):
        assert torch.equal(a, b)
    if cfg["calc_normal"]:
        assert torch.equal(ex_f["normals_volume"], ex_s["normals_volume"]) and torch.equal(ex_f["normals_volume"], ex_c["normals_volume"])
    assert bool(torch.isfinite(rgb_f).all()) and float(rgb_f.min()) >= 0.0 and float(rgb_f.max()) <= 1.0 + 1e-5
    # ... and against the CPU oracle (oracle/render.py, pinned to the reference by the fixtures) on 64 of these rays: HIP vs HIP alone would
    # not notice a configuration both renderers get wrong the same way
    from scipy.spatial import cKDTree
    pick = np.linspace(0, 1998, 64).astype(np.int64)
    orc = common.make_oracle(mesh, state)
    tree = cKDTree(mesh.vertices.astype(np.float64))
    orc.knn_fn = lambda q, v, K: oknn.knn_kdtree(q, v, K, tree=tree)
    ocfg = orender.RenderConfig(N_samples=cfg["N_samples"], N_importance=cfg["N_importance"], N_upsample_iters=cfg["N_upsample_iters"],
                                bounded_near_far=cfg["bounded_near_far"], calc_normal=cfg["calc_normal"], white_bkgd=cfg["white_bkgd"])
    want = orender.render_rays(orc, o[pick].cpu().numpy(), d[pick].cpu().numpy(), ocfg)
    pairs = [("rgb", rgb_f, 1e-4), ("depth_volume", depth_f, 2e-4), ("mask_volume", ex_f["mask_volume"], 1e-4)]
    if cfg["calc_normal"]:
        pairs.append(("normals_volume", ex_f["normals_volume"], 1e-4))
    for key, got, tol in pairs:
        e = np.abs(got[pick].cpu().numpy() - want[key]).reshape(64, -1).max(-1)
        print(f"  {key}: vs oracle on 64 rays: median {np.median(e):.1e}, max {e.max():.2e}, rays beyond {tol:g}: {int((e > tol).sum())}")
        assert np.median(e) <= 2e-6 and (e > tol).sum() <= 1, (key, float(e.max()))   # (one ray: a last-bit sample placement difference)


def _trainer_step_check(cuda_device, torch, backend, fixture):
    from neumesh_amd.trainer import Trainer
    f = common.golden(fixture)
    mesh = common.scene_mesh(int(f["V"]))
    surf = fixture.endswith("_surf")
    model = common.make_model(mesh, common.surface_state(mesh) if surf else common.scene_state(mesh), cuda_device)
    if surf:
        assert abs(float(model.forward_s()) - float(f["s"])) <= 1e-3
    model.autograd_backend = backend
    model.train()
    lw = {str(k): float(v) for k, v in zip(f["loss_weight_keys"], f["loss_weight_vals"])}
    trainer = Trainer(model, loss_weights=lw, teacher_model=None, device_ids=[cuda_device.index or 0])
    trainer.teacher_model = common.StubTeacher()
    H, W = int(f["H"]), int(f["W"])
    args = {"data": {"N_rays": int(f["N_rays"]) if "N_rays" in f.files else 96}}
    kw = dict(N_nograd_samples=2048, N_upsample_iters=4, obj_bounding_radius=1.0, batched=True, perturb=False, white_bkgd=False,
              bounded_near_far=True, calc_normal=True, H=H, W=W, N_samples=64, N_importance=64, rayschunk=4096)
    model_input = {"intrinsics": torch.from_numpy(f["intrinsics"])[None], "c2w": torch.from_numpy(f["c2w"])[None],
                   "object_mask": torch.from_numpy(f["object_mask"])}
    ground_truth = {"rgb": torch.from_numpy(f["gt_rgb"])}
    torch.manual_seed(123)
    ret = trainer.forward(args, None, model_input, ground_truth, kw, 0, device=cuda_device)
    assert np.array_equal(ret["extras"]["select_inds"].cpu().numpy(), f["select_inds"])     # the same random pixels
    def own(key):   # the reference's own change under a 1-ulp nudge of the camera pose (headline-scale fixture on the surface scene)
        return float(f[key]) if key in f.files else 0.0
    for k in ("loss_img", "loss_eikonal", "loss_density", "loss_color", "loss_indicator_vector_reg", "loss_mask", "total"):
        got, want = float(ret["losses"][k]), float(f["loss." + k])
        assert abs(got - want) <= 2e-4 * max(1.0, abs(want)) + 3 * own("self1ulp.loss." + k), (k, got, want)
    assert abs(float(ret["extras"]["psnr"]) - float(f["psnr"])) < 1e-2
    for k in ("xyz", "dirs", "density", "colors", "implicit_nablas", "mask_volume_clipped", "implicit_nablas_norm"):
        assert k in ret["extras"], k
    ret["losses"]["total"].backward()
    checked, worst = 0, {}
    for name, p in model.named_parameters():
        if "grad." + name not in f.files:
            continue
        assert p.grad is not None, name
        g = p.grad.detach().cpu().numpy()
        nrm = float(f["norm." + name])
        # (1 %: scalar parameters such as density_linear.weight_g sum thousands of cancelling per-sample terms)
        rel_n = abs(float(np.linalg.norm(g.astype(np.float64))) - nrm) / max(nrm, 1e-12)
        assert rel_n <= max(1e-2, 3 * own("self1ulp.normrel." + name)) + 1e-6 / max(nrm, 1e-12), (name, rel_n)
        if "rows." + name in f.files:
            g = g[f["rows." + name]]
        # element-wise: 1 % of the tensor's largest entry (the eikonal term differentiates the 2^7-band ds embedding
        # twice: fp32 GPU vs CPU rounding of sin/cos(128 ds) shows up at the 4e-3 level in the first layer's bias)
        rel = np.abs(g - f["grad." + name]).max() / max(np.abs(f["grad." + name]).max(), 1e-8)
        worst[name] = (float(rel), own("self1ulp.gradrel." + name))
        assert rel <= max(1e-2, 3 * own("self1ulp.gradrel." + name)) + 1e-6 / max(np.abs(f["grad." + name]).max(), 1e-8), (name, rel)
        checked += 1
    assert checked >= 20
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["hip", "recompute", "torch"])
def test_trainer_step_matches_reference_trainer(cuda_device, torch_mod, backend):
    """One optimisation step through neumesh_amd.trainer.Trainer.forward (the 2nd element of get_model's tuple,
    called as train.py:176 calls it) against the REFERENCE Trainer on the same scene / camera / ground truth
    (tests/golden/train_step_v3000.npz): the same pixels are drawn, every loss term agrees, and d total / d parameter
    of every model parameter agrees -- with the eikonal term on (second derivative of the nabla graph), the
    distillation terms on (samples_output through the staged renderer) and the mask loss on.  backend: "hip" = the field's forward
    and closed-form backward on the HIP library (_HipField, the default), "recompute" = fused HIP forward + recomputing torch-op
    backward (_FusedField), "torch" = torch ops end to end."""
    _trainer_step_check(cuda_device, torch_mod, backend, "train_step_v3000")


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["hip", "torch"])
def test_trainer_step_headline_scale_surface_scene_matches_reference_trainer(cuda_device, torch_mod, backend):
    """The same step at the scale and on the scene the bench times it on (VERDICT r3 missing #2): V = 140 000, weights of
    synthetic.surface_mlp_state (s = 400), 256 random pixels of a 64 x 64 view -- tests/golden/train_step_v140k_surf.npz, the reference
    Trainer run by oracle/gen_golden.py `train140k`.  With a real surface the sample placement is sensitive to the last bit, so the
    fixture also holds the reference's own change of every loss / gradient under a 1-ulp nudge of the camera pose; a tolerance is the
    v3000 test's or three times that change, whichever is larger (geometry_features' largest gradient entry moves by 18 % in the
    reference itself; losses by < 1e-5)."""
    worst = _trainer_step_check(cuda_device, torch_mod, backend, "train_step_v140k_surf")
    top = sorted(((v[0], v[1], k) for k, v in worst.items()), reverse=True)[:5]
    print("largest relative gradient differences (ours vs reference, reference vs itself + 1 ulp):", [(k, f"{a:.2e}", f"{b:.2e}") for a, b, k in top])


@pytest.mark.gpu
def test_surface_rendering_matches_reference_ray_casting(small, cuda_device, torch_mod):
    """neumesh_amd.ray_casting against the reference's models/ray_casting.py run on the
    reference's field (tests/golden/surface_v3000.npz): first-hit depths / points / masks of
    root_finding_surface_points (256 proposals + 8 secant steps, two level sets) and of
    sphere_tracing_surface_points; then surface_render end to end (colour, depth, normals at the hits)."""
    torch = torch_mod
    from neumesh_amd import ray_casting as rc
    mesh, state, model = small
    f = common.golden("surface_v3000")
    ro = _t(f["rays_o"], cuda_device)[None]
    rd = torch.nn.functional.normalize(_t(f["rays_d"], cuda_device), dim=-1)[None]
    near, far = float(f["near"]), float(f["far"])

    def sdf(p):
        with torch.no_grad():
            return model.forward_density_only(p).squeeze(-1)

    for name in ("tau_a", "tau_b"):
        d, pt, m, msc = rc.root_finding_surface_points(sdf, ro.clone(), rd.clone(), near=near, far=far, batched=True, N_steps=256,
                                                       logit_tau=float(f[name + ".tau"]), method="secant", N_secant_steps=8, fill_inf=False)
        m_ref = f[name + ".mask"]
        # a ray whose bracket value is within the field's parity bound of the level can flip; everything else must agree
        assert (m[0].cpu().numpy() != m_ref).mean() <= 0.03, name
        assert np.array_equal(msc[0].cpu().numpy(), f[name + ".sign_change"]) or (msc[0].cpu().numpy() != f[name + ".sign_change"]).mean() <= 0.03
        both = m[0].cpu().numpy() & m_ref
        assert both.sum() >= 10
        # the field is flat here (|d sdf / d depth| ~ 0.02..0.1), so a 3e-6 field difference moves the root by up to ~1e-4
        np.testing.assert_allclose(d[0].cpu().numpy()[both], f[name + ".d"][both], atol=5e-4)
        np.testing.assert_allclose(pt[0].cpu().numpy()[both], f[name + ".pt"][both], atol=5e-4)
        miss = ~m[0].cpu().numpy() & ~m_ref
        np.testing.assert_allclose(d[0].cpu().numpy()[miss], f[name + ".d"][miss], atol=1e-6)   # `far` or 0 exactly as the reference fills them

        # the early-exit walk (proposals only up to each ray's first sign change) returns what the full evaluation returns
        d_full, pt_full, m_full, msc_full = rc.root_finding_surface_points(sdf, ro.clone(), rd.clone(), near=near, far=far, batched=True, N_steps=256,
                                                                           logit_tau=float(f[name + ".tau"]), method="secant", N_secant_steps=8, fill_inf=False,
                                                                           early_exit=False)
        assert torch.equal(d, d_full) and torch.equal(pt, pt_full) and torch.equal(m, m_full) and torch.equal(msc, msc_full)
        # ... and so does the one-call native form (nm_surface_hits: chained K-NN tiles + geometry MLP over the compacted list of
        # walking rays, bookkeeping in kernels): bit for bit, with scalar and per-ray bounds, inf fill, and without refinement
        surf = rc._NeuMeshSurface(model)
        tau = float(f[name + ".tau"])
        d_n, pt_n, m_n, msc_n = rc.root_finding_surface_points(surf, ro.clone(), rd.clone(), near=near, far=far, batched=True, N_steps=256, logit_tau=tau,
                                                               method="secant", N_secant_steps=8, fill_inf=False)
        # (the native walk forms the proposal positions inside the chained distance kernel: they -- and with them the field values -- can
        #  differ from the torch-op form's in the last bit, so depths agree to ~1e-7 relative, not bit for bit)
        assert torch.equal(m_n, m) and torch.equal(msc_n, msc)
        assert float((d_n - d).abs().max()) <= 2e-6 and float((pt_n - pt).abs().max()) <= 2e-6
        nr = near + 0.1 * torch.rand(ro.shape[:2], device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(4))
        fr = far - 0.2 * torch.rand(ro.shape[:2], device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(5))
        for kw2 in (dict(near=nr, far=fr, fill_inf=True), dict(near=near, far=fr, fill_inf=False, method="none"), dict(near=nr[0], far=far, fill_inf=False, batched=False),
                    dict(near=near, far=far, fill_inf=True, N_steps=100, N_secant_steps=3)):
            kw3 = dict(dict(batched=True, N_steps=256, logit_tau=tau, method="secant", N_secant_steps=8), **kw2)
            o_, d_ = (ro.clone(), rd.clone()) if kw3["batched"] else (ro[0].clone(), rd[0].clone())
            a = rc.root_finding_surface_points(surf, o_.clone(), d_.clone(), **kw3)
            os.environ["NEUMESH_NO_SURFACE_KERNEL"] = "1"
            try:
                b = rc.root_finding_surface_points(surf, o_.clone(), d_.clone(), **kw3)
            finally:
                del os.environ["NEUMESH_NO_SURFACE_KERNEL"]
            for x, y in zip(a, b):
                assert x.shape == y.shape, kw2
                if x.dtype == torch.bool:
                    assert torch.equal(x, y), kw2
                else:
                    fin = torch.isfinite(y)
                    assert torch.equal(torch.isfinite(x), fin) and float((x[fin] - y[fin]).abs().max()) <= 2e-6, kw2

    class Surf:
        def forward(self, p):
            return sdf(p) + 0.08

    d, pt, m = rc.sphere_tracing_surface_points(Surf(), ro.clone(), rd.clone(), near=near, far=far, batched=True, N_iters=20)
    assert np.array_equal(m[0].cpu().numpy(), f["st.mask"])
    np.testing.assert_allclose(d[0].cpu().numpy(), f["st.d"], atol=2e-4)
    # surface_render on the NeuMesh field: colour / nabla at the hit points equal the field queried there
    model_tau = float(f["tau_a.tau"])
    col, dep, ex = rc.surface_render(ro, _t(f["rays_d"], cuda_device)[None], model, calc_normal=True, batched=True, ray_casting_algo="root_finding",
                                     ray_casting_cfgs=dict(near=near, far=far, logit_tau=model_tau, fill_inf=False))
    hit = ex["mask_surface"][0]
    assert tuple(col.shape) == (1, ro.shape[1], 3) and int(hit.sum()) >= 10
    assert float(col[0][~hit].abs().max()) == 0.0 and float(ex["normals_surface"][0][~hit].abs().max()) == 0.0
    pts_hit = (ro[0] + dep[0][:, None] * rd[0])[hit]
    with torch.no_grad():
        sdf_h, rgb_h = model.forward(pts_hit, rd[0][hit])
        _, nab_h = model.forward_with_nablas(pts_hit)
    # the hits sit on the requested level set: exactly for most rays, within the secant method's 8-step residual on the
    # piecewise field (K-NN set changes) for the rest -- the reference's own hits carry the same residual
    res = (sdf_h[:, 0] - model_tau).abs()
    assert float(res.median()) < 1e-5 and float(res.max()) < 0.03
    np.testing.assert_allclose(col[0][hit].cpu().numpy(), rgb_h.cpu().numpy(), atol=2e-5)
    np.testing.assert_allclose(ex["normals_surface"][0][hit].cpu().numpy(), torch.nn.functional.normalize(nab_h, dim=-1).cpu().numpy(), atol=2e-4)
    with pytest.raises(NotImplementedError):
        rc.surface_render(ro, rd, model, ray_casting_algo="")


@pytest.mark.gpu
def test_surface_rendering_headline_scale_surface_scene_matches_reference(surf_scale, cuda_device, torch_mod):
    """Rows a16 / f4 at the scale and on the scene the bench times them on (VERDICT r3 missing #2, weak #3):
    tests/golden/surface_v140k_surf.npz = the reference's root_finding_surface_points + surface_render on the reference NeuMesh field
    of the surface scene at V = 140 000 (512 strided rays of bench frame 0, 324 hit, level 0, 256 proposals + 8 secant steps).
    Gates: hit / sign-change masks EQUAL on every ray whose smallest bracket value exceeds the field's parity bound (the fixture's
    `margin`; the others are counted -- there are none at 1e-5); fill values of missing rays exact; depths / points of the rays on which
    the reference's secant iteration converged (|sdf| at its hit <= 1e-5) within 2e-5, the rest (the iteration wanders on a field that
    jumps where the K-NN set changes) within 2e-3 = a quarter of a bracket; colour at the hits 1e-4, normals 2e-4 on converged rays.
    Both the one-call native form (nm_surface_hits) and the torch-op form over an SDF callable."""
    torch = torch_mod
    from neumesh_amd import ray_casting as rc
    mesh, state, model = surf_scale
    f = common.golden("surface_v140k_surf")
    assert int(f["V"]) == mesh.num_vertices and str(f["state_sha256"]) == common.state_digest(
        {k: v for k, v in state.items() if k not in ("geometry_features", "color_features", "indicator_vector")})
    ro = _t(f["rays_o"], cuda_device)[None]
    rd_raw = _t(f["rays_d"], cuda_device)[None]
    rd = torch.nn.functional.normalize(rd_raw, dim=-1)
    near, far, tau = float(f["near"]), float(f["far"]), float(f["tau"])
    m_ref, sc_ref, d_ref, margin, resid = f["mask"], f["sign_change"], f["d"], f["margin"], f["residual"]
    fragile = margin <= 1e-5
    assert m_ref.sum() >= 300 and (~m_ref).sum() >= 150 and fragile.sum() <= 5

    def sdf(p):
        with torch.no_grad():
            return model.forward_density_only(p).squeeze(-1)

    for label, surf in (("native nm_surface_hits", rc._NeuMeshSurface(model)), ("torch-op walk", sdf)):
        d, pt, m, msc = rc.root_finding_surface_points(surf, ro.clone(), rd.clone(), near=near, far=far, batched=True, N_steps=256, logit_tau=tau,
                                                       method="secant", N_secant_steps=8, fill_inf=False)
        m, msc, d, pt = m[0].cpu().numpy(), msc[0].cpu().numpy(), d[0].cpu().numpy(), pt[0].cpu().numpy()
        assert np.array_equal(m[~fragile], m_ref[~fragile]) and np.array_equal(msc[~fragile], sc_ref[~fragile]), label
        both = m & m_ref
        conv = both & (resid <= 1e-5)
        e = np.abs(d - d_ref)
        print(f"{label}: {int(m.sum())} hits (reference {int(m_ref.sum())}), fragile rays {int(fragile.sum())}, mask flips among them {int((m != m_ref).sum())}; "
              f"depth error on {int(conv.sum())} converged rays: max {e[conv].max():.2e}, on the other {int((both & ~conv).sum())}: max {e[both & ~conv].max():.2e}")
        assert conv.sum() >= 250 and e[conv].max() <= 2e-5, label
        assert e[both & ~conv].max() <= 2e-3, label
        assert np.abs(pt - f["pt"])[conv].max() <= 2e-5
        miss = ~m & ~m_ref
        assert np.array_equal(d[miss], d_ref[miss]), label                      # `far` (or 0 when the ray starts inside), exactly as the reference fills them
    col, dep, ex = rc.surface_render(ro, rd_raw, model, calc_normal=True, batched=True, ray_casting_algo="root_finding",
                                     ray_casting_cfgs=dict(near=near, far=far, logit_tau=tau, fill_inf=False, N_steps=256, N_secant_steps=8))
    hit = ex["mask_surface"][0].cpu().numpy()
    assert np.array_equal(hit[~fragile], m_ref[~fragile])
    conv = hit & m_ref & (resid <= 1e-5)
    col, nrm, nab = col[0].cpu().numpy(), ex["normals_surface"][0].cpu().numpy(), ex["implicit_nablas"][0].cpu().numpy()
    ec, en = np.abs(col - f["color"]).max(-1), np.abs(nrm - f["normals"]).max(-1)
    print(f"surface_render: colour error on converged hits max {ec[conv].max():.2e} (all hits {ec[hit & m_ref].max():.2e}), normals {en[conv].max():.2e}; "
          f"colour std over the hits {f['color'][m_ref].std():.3f}")
    assert ec[conv].max() <= 1e-4 and en[conv].max() <= 2e-4
    assert np.abs(nab - f["nablas"])[conv].max() <= 2e-4 * max(1.0, float(np.abs(f["nablas"][conv]).max()))
    assert float(np.abs(col[~hit]).max()) == 0.0 and float(np.abs(nrm[~hit]).max()) == 0.0 and float(np.abs(f["color"][~m_ref]).max()) == 0.0
    assert f["color"][m_ref].std() > 0.03                                         # a scene with visible colour, not the default-init grey


@pytest.mark.gpu
def test_texture_editable_wrapper_forward_and_render(small, cuda_device, torch_mod):
    """neumesh_amd.editing.TextureEditableNeuMesh (editing/texture_neumesh/texture_neumesh.py:53-122): its fused forward
    equals the reference's formulas evaluated with torch ops on the oracle-checked pieces, unpainted points keep the main
    colour bit for bit, and volume_render drives it through the staged path."""
    torch = torch_mod
    from neumesh_amd.editing import TextureEditableNeuMesh
    from neumesh_amd.renderer import volume_render
    mesh, state, model = small
    V = mesh.num_vertices
    ref_state = dict(state)
    rng = np.random.default_rng(8)
    ref_state["color_features"] = rng.standard_normal((V, 32)).astype(np.float32)
    ref_model = common.make_model(mesh, ref_state, cuda_device)
    mask = torch.from_numpy(mesh.vertices[:, 2] > 0.2).to(cuda_device)[None]          # paint the top cap
    edit_feats = _t(rng.standard_normal((V, 32)).astype(np.float32), cuda_device)
    th = 0.3
    T = torch.tensor([[np.cos(th), -np.sin(th), 0, 0.1], [np.sin(th), np.cos(th), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32,
                     device=cuda_device)
    wrap = TextureEditableNeuMesh(model, [ref_model], mask, edit_feats, T_r_m_list=[T])
    fx = common.golden("field_v3000")
    q, dirs = _t(fx["q"], cuda_device), _t(fx["dirs"], cuda_device)
    with torch.no_grad():
        sdf, rgb = wrap(q, dirs)
        # the reference's formulas, piece by piece, on the same model methods
        s2, nab, ds, idx, w = model.forward(q, dirs, need_nablas=True, nablas_only=True, return_ds=True)
        base = model.forward_color(ds, dirs, model.color_features, indices=idx, weights=w, nabla=nab)
        pm = mask[0][idx]
        pw, uw = (w * pm).sum(-1), (w * (~pm)).sum(-1)
        region = pw > 0
        rw = w * pm
        rw = rw / (rw.sum(-1, keepdim=True) + 1e-8)
        R = T[:3, :3]
        rc_ = ref_model.forward_color(ds[region], (dirs @ R.T)[region], edit_feats, indices=idx[region], weights=rw[region], nabla=(nab @ R.T)[region])
        want = base.clone()
        want[region] = base[region] * (uw / (pw + uw))[region, None] + rc_ * (pw / (pw + uw))[region, None]
    assert torch.equal(sdf, s2)
    assert 0.05 < float(region.float().mean()) < 0.95
    assert torch.equal(rgb[~region], base[~region])                      # untouched where no neighbour is painted
    np.testing.assert_allclose(rgb.cpu().numpy(), want.cpu().numpy(), atol=2e-6)
    assert float((rgb[region] - base[region]).abs().max()) > 1e-3        # ... and the paint is visible
    rf = common.golden("render_v3000_dtu")
    with torch.no_grad():
        img, depth, ex = volume_render(_t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device), wrap, calc_normal=True, perturb=False,
                                       detailed_output=False, rayschunk=4096)
        img0, depth0, _ = volume_render(_t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device), model, calc_normal=True, perturb=False,
                                        detailed_output=False, rayschunk=4096)
    assert torch.equal(depth, depth0) and bool(torch.isfinite(img).all())   # geometry untouched, colours edited
    assert float((img - img0).abs().max()) > 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n_ref,rotated", [(1, False), (2, False), (2, True)])
def test_texture_editing_inside_the_fused_renderer(small, cuda_device, torch_mod, monkeypatch, n_ref, rotated):
    """nm_render_cfg.n_edit: a TextureEditableNeuMesh without a rigid transform is rendered by nm_render_rays itself (painted
    shares, reference colour from the edited table, blend -- editing/texture_neumesh/texture_neumesh.py:79-121) and must
    give the staged renderer's image, which evaluates the wrapper's forward() through the model methods: colours to 2e-6
    (the share sums are the same eight terms in another order), depth / acc / normals bit for bit; one and two references,
    with and without a rigid transform between the models (T_r_m_list: view direction and nabla rotated for the reference)."""
    torch = torch_mod
    from neumesh_amd.editing import TextureEditableNeuMesh
    from neumesh_amd.renderer import fusable_edit_model, make_render_cfg, render_rays_staged, volume_render
    mesh, state, model = small
    V = mesh.num_vertices
    g = torch.Generator().manual_seed(11)
    refs, masks = [], []
    for i in range(n_ref):
        st = dict(state)
        for k in list(st):
            if k.startswith("views_linears") or k.startswith("rgb_linear"):
                st[k] = st[k] + 0.05 * (i + 1) * np.random.default_rng(20 + i).standard_normal(st[k].shape).astype(np.float32)
        refs.append(common.make_model(mesh, st, cuda_device))
        masks.append(torch.rand(V, generator=g) < (0.3 if i == 0 else 0.15))
    feats = (0.1 * torch.randn(V, model.color_features.shape[1], generator=g)).to(cuda_device)
    T_list = None
    if rotated:
        T_list = []
        for i in range(n_ref):
            q, _ = np.linalg.qr(np.random.default_rng(40 + i).standard_normal((3, 3)))
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = q * np.sign(np.linalg.det(q))
            T[:3, 3] = 0.1 * (i + 1)
            T_list.append(torch.from_numpy(T).to(cuda_device))
    wrap = TextureEditableNeuMesh(model, refs, torch.stack(masks).to(cuda_device), feats, T_list)
    assert fusable_edit_model(wrap)
    rf = common.golden("render_v3000_dtu")
    ro, rd = _t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device)
    kw = dict(calc_normal=True, perturb=False, detailed_output=False)
    with torch.no_grad():
        img, depth, ex = volume_render(ro, rd, wrap, rayschunk=4096, **kw)
        img_c, depth_c, _ = volume_render(ro, rd, wrap, rayschunk=17, **kw)
        st_out = render_rays_staged(wrap, ro, rd, make_render_cfg(calc_normal=True), 4096, 1 << 20)
        img0, depth0, _ = volume_render(ro, rd, model, rayschunk=4096, **kw)
        # ... and with three chunks in flight
        monkeypatch.setenv("NEUMESH_RENDER_STREAMS", "3")
        img_o, depth_o, _ = volume_render(ro, rd, wrap, rayschunk=17, **kw)
        monkeypatch.delenv("NEUMESH_RENDER_STREAMS")
    assert torch.equal(img, img_c) and torch.equal(depth, depth_c) and torch.equal(img, img_o) and torch.equal(depth, depth_o)
    assert torch.equal(depth, st_out["depth_volume"]) and torch.equal(ex["mask_volume"], st_out["mask_volume"]) and torch.equal(depth, depth0)
    assert torch.equal(ex["normals_volume"], st_out["normals_volume"])
    np.testing.assert_allclose(img.cpu().numpy(), st_out["rgb"].cpu().numpy(), atol=2e-6)
    assert float((img - img0).abs().max()) > 1e-3 and bool(torch.isfinite(img).all())


@pytest.mark.gpu
def test_device_octree_build_is_bit_identical_to_host_build(cuda_device, torch_mod):
    """nm_grid_create builds the index ON THE DEVICE (nm_grid_build_dev.h: Morton codes, radix sort, per-level node
    kernels); the host build of nm_grid_build.h -- the one tests/hostcheck checks against brute force on the CPU -- is
    the reference implementation.  Node records and sorted vertices must be identical byte for byte: benchmark-scale
    mesh, small meshes, duplicated vertices, forced leaf levels, degenerate clouds (all points equal, points on a line,
    two clusters far apart) and non-finite input."""
    torch = torch_mod
    from neumesh_amd import _lib
    lib = _lib.load_testing()   # host build + export are test hooks of the -DNM_TESTING library (same sources as the product's)
    st = _lib.current_stream(cuda_device)
    rng = np.random.default_rng(3)

    def export(h, V):
        gi = _lib.GridInfo()
        _lib.check(lib.nm_grid_get_info(h, C.byref(gi)), "info")
        nodes = np.empty(gi.num_nodes * 64, np.uint8)
        sv = np.empty((V + 4) * 16, np.uint8)
        _lib.check(lib.nm_grid_debug_export(h, nodes.ctypes.data_as(C.c_void_p), nodes.nbytes, sv.ctypes.data_as(C.c_void_p), sv.nbytes), "export")
        return gi, nodes, sv

    line = np.stack([np.linspace(-1, 1, 5000), np.zeros(5000), np.full(5000, 0.25)], -1)
    clusters = np.concatenate([rng.normal(0, 1e-3, (3000, 3)) - 5.0, rng.normal(0, 1e-3, (3000, 3)) + 5.0])
    cases = [("S-DTU 140k", common.scene_mesh(140000).vertices, 0), ("3000", common.scene_mesh(3000).vertices, 0),
             ("1200 + 64 duplicates", common.scene_mesh(1200, dup=64).vertices, 0), ("20000, leaf level 5", common.scene_mesh(20000).vertices, 5),
             ("20000, leaf level 8", common.scene_mesh(20000).vertices, 8), ("5 vertices", rng.normal(0, 1, (5, 3)), 0),
             ("one vertex", np.array([[0.3, -0.2, 0.9]]), 0), ("all equal", np.tile(np.array([[0.1, 0.2, 0.3]]), (257, 1)), 0),
             ("line", line, 0), ("two far clusters", clusters, 0), ("uniform cloud 100k", rng.uniform(-1, 1, (100000, 3)), 0)]
    for name, verts, level in cases:
        v = _t(np.ascontiguousarray(verts, np.float32), cuda_device)
        V = v.shape[0]
        hd, hh = C.c_void_p(), C.c_void_p()
        _lib.check(lib.nm_grid_create(_lib.ptr(v), V, level, st, C.byref(hd)), "nm_grid_create")
        _lib.check(lib.nm_grid_create_host(_lib.ptr(v), V, level, st, C.byref(hh)), "nm_grid_create_host")
        gd, nd, sd = export(hd, V)
        gh, nh, sh = export(hh, V)
        lib.nm_grid_destroy(hd)
        lib.nm_grid_destroy(hh)
        assert (gd.leaf_level, gd.occupied_leaves, gd.num_nodes, tuple(gd.origin), gd.root_size) == \
               (gh.leaf_level, gh.occupied_leaves, gh.num_nodes, tuple(gh.origin), gh.root_size), name
        assert np.array_equal(sd, sh), name + ": sorted vertices differ"
        assert np.array_equal(nd, nh), name + ": node records differ"
    bad = _t(np.array([[0, 0, 0], [np.nan, 0, 0], [1, 1, 1]], np.float32), cuda_device)
    h = C.c_void_p()
    assert lib.nm_grid_create(_lib.ptr(bad), 3, 0, st, C.byref(h)) != 0 and b"non-finite" in lib.nm_last_error()


@pytest.mark.gpu
def test_small_rayschunk_is_a_lower_bound_by_default(small, cuda_device, torch_mod, monkeypatch):
    """render.py calls the renderer with rayschunk = 4096 (its memory bound).  By default the fused renderer cuts a call into ITS OWN
    chunks (NEUMESH_RAYSCHUNK, default 2^20 rays) because every chunk costs ~26 launch latencies: same pixels bit for bit, one
    nm_render_rays call instead of many; NEUMESH_RAYSCHUNK=0 (what this test suite sets globally) honours the caller's value."""
    torch = torch_mod
    from neumesh_amd import renderer as rmod
    from neumesh_amd import synthetic
    mesh, state, model = small
    H = W = 96
    o, d = synthetic.camera_rays(synthetic.orbit_pose(3), synthetic.pinhole_intrinsics(H, W), H, W)
    o, d = _t(o, cuda_device), _t(d, cuda_device)
    kw = dict(calc_normal=True, perturb=False, detailed_output=False)
    calls = []
    lib = rmod._lib.load()
    real = lib.nm_render_rays

    class Spy:
        def __call__(self, *a):
            calls.append(int(a[5]))
            return real(*a)
    monkeypatch.setattr(lib, "nm_render_rays", Spy())
    with torch.no_grad():
        monkeypatch.setenv("NEUMESH_RAYSCHUNK", "0")
        rgb_a, dep_a, ex_a = rmod.volume_render(o, d, model, rayschunk=1000, **kw)
        n_exact = len(calls)
        monkeypatch.delenv("NEUMESH_RAYSCHUNK")
        rgb_b, dep_b, ex_b = rmod.volume_render(o, d, model, rayschunk=1000, **kw)
    assert n_exact == -(-H * W // 1000) and calls[:n_exact] == [1000] * (n_exact - 1) + [H * W - 1000 * (n_exact - 1)]
    assert calls[n_exact:] == [H * W]
    assert torch.equal(rgb_a, rgb_b) and torch.equal(dep_a, dep_b) and torch.equal(ex_a["normals_volume"], ex_b["normals_volume"])


@pytest.mark.gpu
@pytest.mark.parametrize("over", [dict(W=128, D_density=2, D_color=3), dict(W=64, multires_d=6, multires_view=2), dict(color_dim=96, geometry_dim=72)])
def test_inference_for_configurations_outside_the_fused_kernels(cuda_device, torch_mod, over):
    """The reference takes the hidden width / code widths as free constructor arguments (models/frameworks/neumesh/neumesh.py:16-36);
    the fused inference kernels are tiled for W = 256 and codes <= 64.  A model outside that (VERDICT r3 missing #3) is served under
    no_grad by the any-width kernels of the training path (nm_train_forward, forward only) and rendered by the staged renderer, with
    one warning -- against the CPU oracle built from the model's own state dict: field 3e-6 / 5e-6, rendered frame 1e-4."""
    torch = torch_mod
    import warnings
    from neumesh_amd import MeshGrid, NeuMesh
    from neumesh_amd.renderer import volume_render
    from oracle import field as ofield
    mesh = common.scene_mesh(3000)
    cfg = dict(common.MODEL_CFG, **over)
    torch.manual_seed(5)
    model = NeuMesh(MeshGrid(common.MeshObj(mesh), cuda_device), **cfg).to(cuda_device).eval()
    with torch.no_grad():
        model.indicator_vector.copy_(_t(common.scene_state(mesh)["indicator_vector"], cuda_device))
        model.ln_s.fill_(float(np.log(200.0) / cfg["speed_factor"]))
        model.color_linear[0].weight.mul_(8.0)     # visible colours instead of the default-init grey
    assert not model.fused_supported() and model.train_kernels_supported() and model.inference_route() == "general"
    state = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    ocfg = ofield.FieldConfig(**{k: cfg[k] for k in ("D_density", "D_color", "W", "geometry_dim", "color_dim", "multires_view", "multires_d",
                                                     "multires_fg", "multires_ft", "enable_nablas_input", "speed_factor", "learn_indicator_weight")})
    orc = ofield.OracleField(mesh.vertices, state, ocfg)
    fx, rf = common.golden("field_v3000"), common.golden("render_v3000_dtu")
    near = np.abs(fx["ds"][:, 0]) < 0.5
    q, dirs = fx["q"][near], fx["dirs"][near]
    tq, td = _t(q, cuda_device), _t(dirs, cuda_device)
    model._route_warned = False
    with warnings.catch_warnings(record=True) as rec, torch.no_grad():
        warnings.simplefilter("always")
        sdf0 = model.forward_density_only(tq)
        sdf, nab = model.forward_with_nablas(tq)
        sdf2, rgb, ds, idx, w = model.forward(tq, td, return_ds=True)
        rgb_fc = model.forward_color(ds, td, model.color_features, indices=idx, weights=w, nabla=nab)
    assert sum("outside the fused inference" in str(r.message) for r in rec) == 1          # one warning, not one per call
    o_sdf, o_nab = orc.forward_with_nablas(q)
    _, o_rgb, _ = orc.forward(q, dirs)
    o_ds, o_idx, _ = orc.compute_distance(q)
    assert np.array_equal(idx.cpu().numpy(), o_idx)
    np.testing.assert_allclose(ds.cpu().numpy(), o_ds, atol=3e-6)
    for got in (sdf0, sdf, sdf2):
        np.testing.assert_allclose(got.cpu().numpy(), o_sdf, atol=3e-6)
    assert np.all(np.abs(nab.cpu().numpy() - o_nab) <= 5e-6 + 2e-4 * np.abs(o_ds))
    np.testing.assert_allclose(rgb.cpu().numpy(), o_rgb, atol=5e-6)
    np.testing.assert_allclose(rgb_fc.cpu().numpy(), o_rgb, atol=5e-6)
    assert float(np.std(o_rgb)) > 0.01
    with torch.no_grad():
        img, depth, ex = volume_render(_t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device), model, calc_normal=True, perturb=False,
                                       detailed_output=False, N_samples=64, N_importance=64, rayschunk=4096)
    out = orender.render_rays(orc, rf["rays_o"], rf["rays_d"], orender.RenderConfig(calc_normal=True))
    # (per ray; one ray of the 72 may place a sample on the other side of a crossing -- the last-bit sensitivity of the sampler every
    #  render test of this file allows for; the field gates above are exact)
    n = len(rf["rays_o"])
    for key, got, tol in (("rgb", img, 1e-4), ("depth_volume", depth, 1e-4), ("mask_volume", ex["mask_volume"], 1e-4), ("normals_volume", ex["normals_volume"], 2e-4)):
        e = np.abs(got.cpu().numpy() - out[key]).reshape(n, -1).max(-1)
        assert np.median(e) <= 2e-6 and (e > tol).sum() <= 1, (key, float(e.max()), int((e > tol).sum()))


@pytest.mark.gpu
@pytest.mark.parametrize("path", ["fused", "staged", "autograd", "fused_sampler"])
def test_stochastic_sampler_matches_reference_fixture(small, cuda_device, torch_mod, monkeypatch, path):
    """perturb=True pinned to the reference (tests/golden/render_v3000_perturb.npz: the reference renderer fed a RECORDED sequence of
    uniform numbers in place of torch.rand): the product gets the same numbers -- through nm_render_cfg.u_rand in the fused renderer
    (ABI v9) and the one-call training sampler, through nm_rays_upsample in the staged renderer, with and without autograd -- and must
    place the same samples (sample_pdf(det=False): searchsorted(right=False) on the fp32 CDF, the 1e-5 guards) and render the same frame."""
    torch = torch_mod
    from neumesh_amd import renderer as rmod
    mesh, state, _ = small
    model = common.make_model(mesh, state, cuda_device)
    f, rf = common.golden("render_v3000_perturb"), common.golden("render_v3000_dtu")
    u = torch.from_numpy(f["u"]).to(cuda_device)            # [4, R, 16]
    calls = []

    class FakeTorch:
        """neumesh_amd.renderer's `torch` with rand() replaced: a [R, n] request is the next iteration's block, a [iters, R, n] request all"""
        def __getattr__(self, name):
            return getattr(torch, name)

        def rand(self, shape, **kw):
            shape = tuple(shape)
            if len(shape) == 3:
                calls.append(shape)
                return u.clone()
            blk = u[len(calls) % u.shape[0]].clone()
            calls.append(shape)
            assert tuple(blk.shape) == shape
            return blk
    monkeypatch.setattr(rmod, "torch", FakeTorch())
    ro, rd = _t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device)
    kw = dict(calc_normal=True, N_samples=64, N_importance=64, perturb=True, detailed_output=True, rayschunk=4096)
    if path == "fused":
        with torch.no_grad():
            rgb, depth, ex = rmod.volume_render(ro, rd, model, **kw)
    elif path == "staged":
        with torch.no_grad():
            ex = rmod.render_rays_staged(model, ro, rd, rmod.make_render_cfg(calc_normal=True), 4096, 1 << 20, detailed=True, perturb=True)
            rgb, depth = ex["rgb"], ex["depth_volume"]
    else:
        monkeypatch.setenv("NEUMESH_FUSED_SAMPLER", "1" if path == "fused_sampler" else "0")
        model.train()
        rgb, depth, ex = rmod.volume_render(ro, rd, model, **kw)
        assert rgb.requires_grad
    assert len(calls) == (1 if path in ("fused", "fused_sampler") else 4)
    g = {k: v.detach().cpu().numpy() for k, v in ex.items() if torch.is_tensor(v)}
    np.testing.assert_allclose(rgb.detach().cpu().numpy(), f["rgb"], atol=1e-4)
    np.testing.assert_allclose(depth.detach().cpu().numpy(), f["depth_volume"], atol=1e-4)
    np.testing.assert_allclose(g["mask_volume"], f["mask_volume"], atol=1e-4)
    np.testing.assert_allclose(g["normals_volume"], f["normals_volume"], atol=2e-4 if path in ("autograd", "fused_sampler") else 1e-4)
    d_mid_ref = 0.5 * (f["d_all"][:, 1:] + f["d_all"][:, :-1])
    dd = np.abs(g["d_final"] - d_mid_ref)                                          # the reference's sample placement (a uniform number within an
    assert (dd <= 2e-6).mean() >= 0.995 and dd.max() < 5e-3, (float(dd.max()), float((dd > 2e-6).mean()))   # ulp of a CDF edge may change bins)
    assert np.abs(f["d_all"] - rf["d_all"]).max() > 1e-3                           # ... which is not the deterministic one


# --------------------------------------------------------------------- several chunks in flight
@pytest.mark.gpu
@pytest.mark.parametrize("lanes", [2, 3, 4])
def test_chunks_in_flight_render_identical_pixels(surf_scale, cuda_device, torch_mod, monkeypatch, lanes):
    """A call cut into ray chunks on several streams (each lane with its own workspace) returns every output bit for bit as the same call
    rendered in one piece; also with a ragged last chunk, normals off and the lego shape.  (Round 5's pull-form K-NN kernels that yielded
    their SIMDs to the other chunks' MLP launches -- nm_render_cfg.overlap, ABI v10 -- were a measured loss and left the library in round 6.)"""
    torch = torch_mod
    from neumesh_amd import renderer as rmod
    from neumesh_amd import synthetic
    mesh, state, model = surf_scale
    H = W = 192
    o, d = synthetic.camera_rays(synthetic.orbit_pose(2), synthetic.pinhole_intrinsics(H, W), H, W)
    o, d = _t(o, cuda_device), _t(d, cuda_device)
    for kw in (dict(calc_normal=True), dict(calc_normal=False, N_samples=32, N_importance=32, white_bkgd=True)):
        kw.update(perturb=False, detailed_output=False)
        monkeypatch.setenv("NEUMESH_RAYSCHUNK", "0")
        monkeypatch.setenv("NEUMESH_RENDER_STREAMS", "1")
        with torch.no_grad():
            rgb_a, dep_a, ex_a = rmod.volume_render(o, d, model, rayschunk=H * W, **kw)
        monkeypatch.setenv("NEUMESH_RENDER_STREAMS", str(lanes))
        with torch.no_grad():
            rgb_b, dep_b, ex_b = rmod.volume_render(o, d, model, rayschunk=5000, **kw)
        torch.cuda.synchronize()
        assert torch.equal(rgb_a, rgb_b) and torch.equal(dep_a, dep_b) and torch.equal(ex_a["mask_volume"], ex_b["mask_volume"])
        if kw["calc_normal"]:
            assert torch.equal(ex_a["normals_volume"], ex_b["normals_volume"])


@pytest.mark.gpu
def test_mid_point_sub_passes_render_identical_pixels(small, surf_scale, cuda_device, torch_mod, monkeypatch):
    """nm_render_cfg.mid_passes (ABI v11): the mid-point stage in 2 / 4 / 5 sub-passes over ray ranges that share one record
    region returns every output bit for bit as the one-pass layout of round 5 -- plain model (headline shape and the lego shape), the detailed
    call, and a texture-edited model with rotated references (the blend's arrays are list-addressed too) -- while the workspace shrinks."""
    torch = torch_mod
    from neumesh_amd import _lib, synthetic
    from neumesh_amd import renderer as rmod
    lib = _lib.load()
    mesh, state, model = surf_scale
    H = W = 368                                   # 135 424 rays: four sub-passes of >= 32 768 rays
    o, d = synthetic.camera_rays(synthetic.orbit_pose(3), synthetic.pinhole_intrinsics(H, W), H, W)
    o, d = _t(o, cuda_device), _t(d, cuda_device)
    monkeypatch.setenv("NEUMESH_RAYSCHUNK", "0")
    monkeypatch.setenv("NEUMESH_RENDER_STREAMS", "1")
    for kw in (dict(calc_normal=True), dict(calc_normal=False, N_samples=32, N_importance=32, white_bkgd=True), dict(calc_normal=True, detailed_output=True)):
        kw = dict(dict(perturb=False, detailed_output=False), **kw)
        outs = {}
        for q in (1, 2, 4, 5):
            monkeypatch.setenv("NEUMESH_MID_PASSES", str(q))
            with torch.no_grad():
                outs[q] = rmod.volume_render(o, d, model, rayschunk=H * W, **kw)
        torch.cuda.synchronize()
        for q in (2, 4, 5):
            assert torch.equal(outs[1][0], outs[q][0]) and torch.equal(outs[1][1], outs[q][1]), (q, kw)
            for k in ("mask_volume", "normals_volume", "radiance", "implicit_nablas"):
                if k in outs[1][2]:
                    assert torch.equal(outs[1][2][k], outs[q][2][k]), (q, k)
    # a ragged ray count (70 225 rays: two sub-passes of 35 136 and 35 089 rays, the last depth-bucket group partly filled)
    Hr = Wr = 265
    o2, d2 = synthetic.camera_rays(synthetic.orbit_pose(9), synthetic.pinhole_intrinsics(Hr, Wr), Hr, Wr)
    o2, d2 = _t(o2, cuda_device), _t(d2, cuda_device)
    outs = {}
    for q in (1, 2):
        monkeypatch.setenv("NEUMESH_MID_PASSES", str(q))
        with torch.no_grad():
            outs[q] = rmod.volume_render(o2, d2, model, rayschunk=Hr * Wr, calc_normal=True, perturb=False, detailed_output=False)
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1]) and torch.equal(outs[1][2]["normals_volume"], outs[2][2]["normals_volume"])
    ws = {}
    for q in (1, 0):
        cfg = rmod.make_render_cfg(calc_normal=True, mid_passes=q)
        cfg.code_dims = 32 | (32 << 16)
        ws[q] = int(lib.nm_render_workspace_bytes(C.byref(cfg), 327680))
    print(f"workspace of a 327 680-ray chunk: one pass {ws[1] / 1e9:.2f} GB ({ws[1] / 327680 / 1024:.1f} KiB per ray), default {ws[0] / 1e9:.2f} GB ({ws[0] / 327680 / 1024:.1f} KiB per ray)")
    assert ws[0] <= 13.5e9 < ws[1]
    # texture editing through the sub-passes (two rotated references at V = 3000; 65 536 rays = two sub-passes)
    mesh3, state3, _ = small
    wrap, _main = common.edit_model(mesh3, state3, 2, True, cuda_device)
    assert rmod.fusable_edit_model(wrap)
    H = W = 256
    o, d = synthetic.camera_rays(synthetic.orbit_pose(7), synthetic.pinhole_intrinsics(H, W), H, W)
    o, d = _t(o, cuda_device), _t(d, cuda_device)
    outs = {}
    for q in (1, 2):
        monkeypatch.setenv("NEUMESH_MID_PASSES", str(q))
        with torch.no_grad():
            outs[q] = rmod.volume_render(o, d, wrap, rayschunk=H * W, calc_normal=True, perturb=False, detailed_output=False)
    assert torch.equal(outs[1][0], outs[2][0]) and torch.equal(outs[1][1], outs[2][1]) and torch.equal(outs[1][2]["normals_volume"], outs[2][2]["normals_volume"])


@pytest.mark.gpu
def test_profile_clock_reads_a_plausible_shader_clock(cuda_device, torch_mod):
    """nm_profile_clock (instrumentation behind bench.py's roofline.shader_clock_mhz_under_load): one wave counts its cycle counter against the
    constant 100 MHz counter; idle or busy, an MI355X answers between a few hundred MHz and its 2.4 GHz boost."""
    torch = torch_mod
    from neumesh_amd import _lib
    lib = _lib.load()
    mhz = C.c_float()
    for _ in range(3):
        _lib.check(lib.nm_profile_clock(200, C.byref(mhz), _lib.current_stream(cuda_device)), "nm_profile_clock")
        assert 100.0 < mhz.value < 3500.0, mhz.value
    assert lib.nm_profile_clock(0, C.byref(mhz), _lib.current_stream(cuda_device)) != 0     # bad arguments are refused


# --------------------------------------------------------------------- fp16-range fall-back of the split-half MLP modes
@pytest.mark.gpu
@pytest.mark.parametrize("precision", ["f16x2s", "f16x2"])
def test_fp16_range_overflow_falls_back_to_fp32(small, cuda_device, torch_mod, precision):
    """A weight set whose activations leave the fp16 range (a checkpoint the split-half format does not fit): the kernels raise the device flag,
    the renderer notices after the call, warns, switches the model to the fp32 kernels and renders the call AGAIN -- what the caller gets is the
    fp32 render bit for bit, never the Inf / NaN-affected first attempt.  Here: the second colour layer's weight-norm gain x 3e5 (the ReLU
    activations behind it reach ~1e6; the sigmoid output stays finite in fp32)."""
    import warnings
    torch = torch_mod
    from neumesh_amd import renderer as rmod
    from neumesh_amd import synthetic
    mesh, state, _ = small
    model = common.make_model(mesh, state, cuda_device)     # (its own model: the weights are edited below)
    H = W = 48
    o, d = synthetic.camera_rays(synthetic.orbit_pose(5), synthetic.pinhole_intrinsics(H, W), H, W)
    o, d = _t(o, cuda_device), _t(d, cuda_device)
    kw = dict(calc_normal=True, perturb=False, detailed_output=False, rayschunk=H * W)
    lin = [m for m in model.modules() if hasattr(m, "weight_g") and m.weight_g.shape[0] == 256][-1]    # the last 256-wide weight-normed layer: colour network
    with torch.no_grad():
        lin.weight_g.mul_(3e5)
    model.mlp_precision = "fp32"
    with torch.no_grad():
        rgb_ref, dep_ref, ex_ref = rmod.volume_render(o, d, model, **kw)
    assert bool(torch.isfinite(rgb_ref).all())
    model.mlp_precision = precision
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with torch.no_grad():
            rgb, dep, ex = rmod.volume_render(o, d, model, **kw)
    assert any("fp16 range" in str(w.message) for w in caught), [str(w.message)[:80] for w in caught]
    assert model.mlp_precision == "fp32"                                   # the model stays on the fp32 kernels from here on
    assert torch.equal(rgb, rgb_ref) and torch.equal(dep, dep_ref) and torch.equal(ex["normals_volume"], ex_ref["normals_volume"])
    model.mlp_precision = precision                                        # point-wise calls notice too
    xyz = _t(np.asarray(mesh.vertices[:512], np.float32) * 1.01, cuda_device)
    dirs = torch.nn.functional.normalize(torch.ones_like(xyz), dim=-1)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with torch.no_grad():
            model.forward(xyz, dirs)
            model.check_fp16_range(force=True)
    assert model.mlp_precision == "fp32" and any("fp16 range" in str(w.message) for w in caught)


@pytest.mark.gpu
def test_render_call_returns_without_host_sync_and_reports_a_late_overflow(small, cuda_device, torch_mod, monkeypatch):
    """SURVEY 8b "no hidden sync" (VERDICT r5 weak #6 / item 5).  The first fused call on a weight set reads the fp16-range flag with a stream
    sync (and would re-render in fp32 before returning); every later call only POSTS an asynchronous read (nm_field_overflow_post) and returns
    while its kernels are still running.  An overflow in such a call -- here caused by the INPUTS: a colour-code table scaled by 1e6, which does
    not re-pack the weights -- is found at the next entry: a warning that names the earlier call as invalid, the model on the fp32 kernels from
    there on, the new call's pixels equal to the fp32 render.  NEUMESH_EAGER_RANGE_CHECK=1 restores check-and-re-render inside every call."""
    import warnings
    torch = torch_mod
    from neumesh_amd import renderer as rmod
    from neumesh_amd import synthetic
    if common.DEFAULT_PRECISION == "fp32":
        pytest.skip("the fp32 kernels have no fp16-range flag to read")
    mesh, state, _ = small
    model = common.make_model(mesh, state, cuda_device)
    H = W = 256
    o, d = synthetic.camera_rays(synthetic.orbit_pose(5), synthetic.pinhole_intrinsics(H, W), H, W)
    o, d = _t(o, cuda_device), _t(d, cuda_device)
    kw = dict(calc_normal=True, perturb=False, detailed_output=False, rayschunk=H * W)
    monkeypatch.delenv("NEUMESH_EAGER_RANGE_CHECK", raising=False)
    with torch.no_grad():
        rgb0, _, _ = rmod.volume_render(o, d, model, **kw)          # first call on this weight set: eager
        assert model._range_checked and not model._range_pending
        torch.cuda.synchronize()
        rgb1, _, _ = rmod.volume_render(o, d, model, **kw)          # deferred
        still_running = not torch.cuda.current_stream(cuda_device).query()
        assert len(model._range_pending) == 1
        assert model.synchronize_fp16_range() and not model._range_pending
        rgb1b, _, _ = rmod.SingleRenderer(model)(o[None], d[None], batched=True, **kw)
        assert len(model._range_pending) == 1 and rmod.SingleRenderer(model).synchronize() and not model._range_pending
        assert torch.equal(rgb1b[0], rgb1)
    assert still_running, "the deferred call returned only after its kernels had finished"
    assert torch.equal(rgb0, rgb1) and model.mlp_precision == common.DEFAULT_PRECISION
    with torch.no_grad():
        model.color_features.data.mul_(1e6)                           # inputs outside the fp16 range; the packed weights are untouched
        bad, _, _ = rmod.volume_render(o, d, model, **kw)           # deferred: returns whatever the f16 kernels made of it
        assert model.mlp_precision == common.DEFAULT_PRECISION and len(model._range_pending) == 1
        torch.cuda.synchronize()
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            rgb2, _, _ = rmod.volume_render(o, d, model, **kw)      # entry poll finds the flag of the EARLIER call
        assert any("EARLIER render call" in str(w.message) for w in caught), [str(w.message)[:120] for w in caught]
        assert model.mlp_precision == "fp32"
        ref, _, _ = rmod.volume_render(o, d, model, **kw)
        assert torch.equal(rgb2, ref) and bool(torch.isfinite(rgb2).all())
        # eager on request: the overflowing call itself is repeated by the fp32 kernels
        model.mlp_precision = common.DEFAULT_PRECISION
        monkeypatch.setenv("NEUMESH_EAGER_RANGE_CHECK", "1")
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            rgb3, _, _ = rmod.volume_render(o, d, model, **kw)
        assert any("re-running this call" in str(w.message) for w in caught) and model.mlp_precision == "fp32"
        assert torch.equal(rgb3, ref)
