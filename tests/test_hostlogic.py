"""CPU: the host/device-shared headers of neumesh_amd/csrc (octree K-NN traversal, projected
distance + closed-form gradient, per-ray stages) compiled with g++ (tests/hostcheck) and checked
against the oracle.  This pins the ALGORITHMS the device kernels run; the kernels themselves are
checked on the GPU (tests marked gpu)."""
import numpy as np
import pytest

import common
from hostcheck.loader import HostGrid, P, load
from oracle import knn as oknn, render as orender


def _queries(verts, n, seed):
    rng = np.random.default_rng(seed)
    V = len(verts)
    k = min(8, V)
    return np.concatenate([
        verts[rng.integers(0, V, n // 2)] + 0.01 * rng.standard_normal((n // 2, 3)),   # near surface
        verts[rng.integers(0, V, n // 4)] + 0.2 * rng.standard_normal((n // 4, 3)),    # mid range
        rng.uniform(-3, 3, (n - n // 2 - n // 4 - k, 3)),                              # far / outside bbox
        verts[:k],                                                                      # exactly on vertices
    ]).astype(np.float32)


@pytest.mark.parametrize("V,dup,K,level", [(3000, 0, 8, 0), (1200, 64, 8, 0), (20000, 0, 8, 0), (5000, 0, 1, 0),
                                           (5000, 0, 16, 0), (5000, 0, 32, 0), (3000, 0, 8, 2), (3000, 0, 8, 7),
                                           (9, 0, 8, 0), (5, 0, 8, 0), (1, 0, 8, 0)])
def test_octree_knn_is_bit_exact(V, dup, K, level):
    verts = common.scene_mesh(V, dup).vertices
    q = _queries(verts, 2048 if V > 100 else 64, V + K)
    idx, d2 = HostGrid(verts, level).knn(q, K)
    ridx, rd2 = oknn.knn_bruteforce(q, verts, K)
    assert np.array_equal(idx, ridx)
    assert np.array_equal(d2, rd2)


def test_octree_knn_dtu_scale_mesh():
    verts = common.scene_mesh(140000).vertices
    q = _queries(verts, 6000, 7)
    g = HostGrid(verts)
    assert 5 <= g.level <= 8
    idx, d2 = g.knn(q, 8)
    ridx, rd2 = oknn.knn_bruteforce(q, verts, 8)
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


def test_octree_knn_degenerate_clouds():
    rng = np.random.default_rng(3)
    line = np.zeros((500, 3), np.float32); line[:, 0] = np.linspace(-1, 1, 500)
    same = np.tile(np.array([[0.3, -0.2, 0.1]], np.float32), (40, 1))
    clustered = np.concatenate([0.001 * rng.standard_normal((400, 3)), 5 + 0.001 * rng.standard_normal((400, 3))]).astype(np.float32)
    for verts in (line, same, clustered):
        q = np.concatenate([verts[:50], rng.uniform(-6, 6, (300, 3)).astype(np.float32)])
        idx, d2 = HostGrid(verts).knn(q, 8)
        ridx, rd2 = oknn.knn_bruteforce(q, verts, 8)
        assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)


def test_projected_distance_and_gradient_match_fixture():
    fx = common.golden("field_v3000")
    mesh = common.scene_mesh(3000)
    st = common.scene_state(mesh)
    ds, idx, w, g = HostGrid(mesh.vertices).compute_distance(fx["q"], st["indicator_vector"], 0.1)
    assert np.array_equal(idx, fx["idx"])
    np.testing.assert_allclose(ds, fx["ds"][:, 0], atol=2e-6)
    np.testing.assert_allclose(w, fx["w"], atol=2e-6)
    np.testing.assert_allclose(g, fx["dds_dx"], atol=2e-5, rtol=2e-5)


def test_linspace_matches_torch_formula():
    lib = load()
    for n in (2, 16, 64, 256):
        out = np.empty(n, np.float32)
        lib.hc_linspace01(n, P(out))
        assert np.array_equal(out, orender.torch_linspace01(n))


def test_ray_setup_and_bounds():
    lib = load()
    rf = common.golden("render_v3000_dtu")
    R = len(rf["rays_o"])
    dirn, nf0 = np.empty((R, 3), np.float32), np.empty((R, 2), np.float32)
    lib.hc_ray_setup(P(np.ascontiguousarray(rf["rays_o"])), P(np.ascontiguousarray(rf["rays_d"])), R, 1.0, P(dirn), P(nf0))
    odir = orender.normalize(rf["rays_d"])
    on, of = orender.near_far_from_sphere(rf["rays_o"], odir, 1.0)
    np.testing.assert_allclose(dirn, odir, atol=1e-7)
    np.testing.assert_allclose(nf0, np.concatenate([on, of], 1), atol=1e-6)
    # bounds: feed the oracle's probe distances
    mesh = common.scene_mesh(3000)
    orc = common.make_oracle(mesh, common.scene_state(mesh))
    n2, f2, probe = orender.compute_bounded_near_far(orc, rf["rays_o"], odir, on, of)
    nf = np.empty((R, 2), np.float32)
    lib.hc_ray_bounds(P(np.ascontiguousarray(probe, np.float32)), R, 256, 0.1, P(np.concatenate([on, of], 1).astype(np.float32)), P(nf))
    np.testing.assert_allclose(nf, np.concatenate([n2, f2], 1), atol=1e-6)
    np.testing.assert_allclose(nf, np.concatenate([rf["near"], rf["far"]], 1), atol=1e-6)


def test_upsample_merge_follow_the_oracle():
    """Drive the C++ per-ray up-sampling with the ORACLE's field values and compare every
    iteration's new depths / the final sorted depth list."""
    lib = load()
    rf = common.golden("render_v3000_dtu")
    mesh = common.scene_mesh(3000)
    orc = common.make_oracle(mesh, common.scene_state(mesh))
    R, cap = len(rf["rays_o"]), 128
    odir = orender.normalize(rf["rays_d"])
    d = np.zeros((R, cap), np.float32); sdf = np.zeros((R, cap), np.float32)
    d[:, :64], sdf[:, :64] = rf["d_coarse"], rf["sdf_coarse"]
    od, osdf = rf["d_coarse"].copy(), rf["sdf_coarse"].copy()
    n, pending = 64, 0
    for it in range(4):
        if pending:
            lib.hc_ray_merge(P(d), P(sdf), R, cap, n - pending, pending)
        lib.hc_ray_upsample(P(d), P(sdf), R, cap, n, it, 16)
        o_fine, _ = orender.upsample_step(od, osdf, it, 16)
        got = d[:, n:n + 16]
        # all but the u=1 sample agree to rounding; the last one is placed by whether the fp32
        # cdf[-1] rounded above or below 1.0 (oracle/compare.py): anywhere inside the last bin
        np.testing.assert_allclose(got[:, :15], o_fine[:, :15], atol=3e-6)
        assert ((got[:, 15] >= od[:, -2] - 3e-6) & (got[:, 15] <= od[:, -1] + 3e-6)).all()
        # continue both sides from the ORACLE's samples so the comparison stays aligned
        d[:, n:n + 16] = o_fine
        pts = (rf["rays_o"][:, None, :] + o_fine[..., None] * odir[:, None, :]).astype(np.float32)
        s_f = orc.forward_density_only(pts)[..., 0]
        sdf[:, n:n + 16] = s_f
        od = np.concatenate([od, o_fine], -1); osdf = np.concatenate([osdf, s_f], -1)
        order = np.argsort(od, -1, kind="stable")
        od, osdf = np.take_along_axis(od, order, -1), np.take_along_axis(osdf, order, -1)
        n += 16; pending = 16
    lib.hc_ray_merge(P(d), P(sdf), R, cap, n - pending, pending)
    assert np.array_equal(d, od) and np.array_equal(sdf, osdf)


def test_composite_matches_reference_fixture():
    lib = load()
    rf = common.golden("render_v3000_dtu")
    R, N = rf["implicit_surface"].shape
    d_all = np.ascontiguousarray(rf["d_all"], np.float32)
    # the reference's own per-sample outputs in, its composited pixels out
    rgb, depth, acc, nrm = np.empty((R, 3), np.float32), np.empty(R, np.float32), np.empty(R, np.float32), np.empty((R, 3), np.float32)
    fx = common.golden("field_v3000")
    lib.hc_ray_composite(P(np.ascontiguousarray(rf["implicit_surface"])), P(d_all), R, N, float(fx["s"]),
                         P(np.ascontiguousarray(rf["radiance"])), P(np.ascontiguousarray(rf["implicit_nablas"])), 0,
                         P(rgb), P(depth), P(acc), P(nrm))
    np.testing.assert_allclose(rgb, rf["rgb"], atol=2e-6)
    np.testing.assert_allclose(acc, rf["mask_volume"], atol=2e-6)
    np.testing.assert_allclose(depth, rf["depth_volume"], atol=2e-5)   # d_all is recovered to ~1e-7 only
    np.testing.assert_allclose(nrm, rf["normals_volume"], atol=2e-6)


def test_fast_softplus_formula():
    """numpy emulation of nm_softplus100 (neumesh_amd/csrc/nm_mlp.h): exp2/log2-based softplus and
    derivative vs float64 truth -- same error class as the libm fp32 form torch uses."""
    from oracle.field import softplus100, softplus100_grad
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-0.3, 0.3, 400000), rng.uniform(-1e-3, 1e-3, 50000), rng.uniform(0.19, 0.21, 5000),
                        np.array([-10.0, -1.0, 0.0, 0.2, 0.2000001, 5.0])]).astype(np.float32)
    f = np.float32
    z = np.exp2(np.minimum(x * f(144.269504), f(30.2965958))).astype(np.float32)
    u = f(1) + z
    y = np.maximum(x, np.log2(u).astype(np.float32) * f(0.0069314718)).astype(np.float32)
    g = (z / u).astype(np.float32)
    xd = x.astype(np.float64)
    with np.errstate(over="ignore"):
        yt = np.where(xd * 100 > 20, xd, np.log1p(np.exp(xd * 100)) / 100)
        gt = np.where(xd * 100 > 20, 1.0, 1 / (1 + np.exp(-xd * 100)))
    assert np.abs(y - yt).max() < 6e-8 and np.abs(g - gt).max() < 3e-7
    assert np.abs(y - softplus100(x)).max() < 6e-8 and np.abs(g - softplus100_grad(x)).max() < 3e-7


def test_warm_started_search_is_exact_and_cheaper():
    """Warm start: bound = (distance of a neighbouring sample to its 8th vertex) + (gap between the two
    samples) is a valid upper bound, so the warm search returns the same K-NN, visiting fewer nodes."""
    mesh = common.scene_mesh(20000)
    g = HostGrid(mesh.vertices)
    rng = np.random.default_rng(4)
    o = rng.uniform(-0.2, 0.2, (64, 3)).astype(np.float32) + np.array([2.0, 0, 0], np.float32)
    dirs = orender.normalize((-o + 0.3 * rng.standard_normal((64, 3))).astype(np.float32))
    t = np.sort(rng.uniform(1.0, 3.0, (64, 40)).astype(np.float32), axis=1)
    pts = (o[:, None, :] + t[..., None] * dirs[:, None, :]).astype(np.float32)
    ridx, rd2 = oknn.knn_bruteforce(pts.reshape(-1, 3), mesh.vertices, 8)
    R = np.sqrt(rd2[:, 7]).reshape(64, 40)
    gap = np.linalg.norm(pts[:, 1:] - pts[:, :-1], axis=-1)
    bound = np.concatenate([np.full((64, 1), 1e9, np.float32), (R[:, :-1] + gap).astype(np.float32)], 1)   # from the previous sample
    idx, d2, n_nodes, n_verts = g.knn_warm(pts.reshape(-1, 3), bound.reshape(-1))
    assert np.array_equal(idx, ridx) and np.array_equal(d2, rd2)
    cold = g.knn_stats(pts.reshape(-1, 3))
    assert n_nodes < cold[0] and n_verts < 0.95 * cold[1]   # saves vertex visits / top-K insertions (leaves hold ~30 vertices)


def test_upsample_with_random_u_follows_the_oracle():
    """sample_pdf(det=False) (perturb=True, rend_util.py:293-296): the C++ stage with the caller's
    uniform randoms vs the oracle's restatement with the same u (unordered u: binary lower bound)."""
    lib = load()
    rf = common.golden("render_v3000_dtu")
    R, cap = len(rf["rays_o"]), 128
    d = np.zeros((R, cap), np.float32); sdf = np.zeros((R, cap), np.float32)
    d[:, :64], sdf[:, :64] = rf["d_coarse"], rf["sdf_coarse"]
    rng = np.random.default_rng(21)
    for it in range(3):
        u = rng.random((R, 16), dtype=np.float32)
        u[:, 0] = 0.0   # torch.rand's lower end ([0, 1)); in the flat tail of the cdf (increments of ~1e-7 per bin, u
        #                beyond ~0.999 here) bin choice hinges on the last bit of the running sum, see oracle/compare.py
        lib.hc_ray_upsample_u(P(d), P(sdf), R, cap, 64, it, 16, P(u))
        o_fine, _ = orender.upsample_step(rf["d_coarse"], rf["sdf_coarse"], it, 16, u=u)
        np.testing.assert_allclose(d[:, 64:80], o_fine, atol=3e-6)
        assert (d[:, 64:80] >= rf["d_coarse"][:, :1] - 1e-6).all() and (d[:, 64:80] <= rf["d_coarse"][:, -1:] + 1e-6).all()


def test_upsample_slot_tracking_and_bounds():
    """nm_rays_upsample_kernel's bookkeeping: slot[j] always names the generation position of the
    sample now at sorted position j, and the emitted warm-start bounds are valid upper bounds."""
    import ctypes as C
    lib = load()
    rf = common.golden("render_v3000_dtu")
    mesh = common.scene_mesh(3000)
    orc = common.make_oracle(mesh, common.scene_state(mesh))
    R, cap = len(rf["rays_o"]), 128
    odir = orender.normalize(rf["rays_d"])
    d = np.zeros((R, cap), np.float32); sdf = np.zeros((R, cap), np.float32)
    slot = np.zeros((R, cap), np.int32); radius = np.zeros((R, cap), np.float32); bound = np.zeros((R, cap), np.float32)
    gen_d = np.zeros((R, cap), np.float32)      # depth by generation position
    d[:, :64], sdf[:, :64] = rf["d_coarse"], rf["sdf_coarse"]
    gen_d[:, :64] = rf["d_coarse"]

    def true_radius(dep):
        pts = (rf["rays_o"][:, None, :] + dep[..., None] * odir[:, None, :]).astype(np.float32)
        _, d2 = oknn.knn_bruteforce(pts.reshape(-1, 3), mesh.vertices, 8)
        return np.sqrt(d2[:, 7]).reshape(dep.shape), pts
    radius[:, :64], _ = true_radius(rf["d_coarse"])
    n, pending = 64, 0
    i32p = C.POINTER(C.c_int32)
    for it in range(4):
        lib.hc_ray_upsample_slots(P(d), P(sdf), slot.ctypes.data_as(i32p), P(radius), P(bound), R, cap, n, pending, it, 16)
        assert np.array_equal(np.take_along_axis(gen_d, slot[:, :n].astype(np.int64), 1), d[:, :n])     # slots consistent
        assert np.all(np.diff(d[:, :n], axis=1) >= 0)
        new = d[:, n:n + 16].copy()
        gen_d[:, n:n + 16] = new
        rad_new, pts = true_radius(new)
        assert np.all(bound[:, n:n + 16] * 1.0001 + 1e-5 >= rad_new)                                   # valid upper bounds
        radius[:, n:n + 16] = rad_new
        sdf[:, n:n + 16] = orc.forward_density_only(pts)[..., 0]
        n += 16; pending = 16


def test_fast_sincos_formula():
    """numpy emulation of nm_sincos (neumesh_amd/csrc/nm_mlp.h): 3-term Cody-Waite reduction by pi/2
    + minimax polynomials; same error class as libm's fp32 sin/cos over the encodings' range."""
    f32 = np.float32

    def fma(a, b, c):
        return (np.asarray(a, np.float64) * np.float64(b) + np.asarray(c, np.float64)).astype(np.float32)

    def sincos(x):
        k = np.rint((x * f32(0.63661977236758134)).astype(np.float32)).astype(np.float32)
        r = fma(k, f32(-1.57079637050628662109e+00), x)
        r = fma(k, f32(4.37113882867379288655e-08), r)
        r = fma(k, f32(1.71512451000588187280e-15), r)
        q = k.astype(np.int64)
        r2 = (r * r).astype(np.float32)
        ps = fma(r2, f32(-1.9515295891e-4), np.full_like(r2, f32(8.3321608736e-3)))
        ps = fma(ps, r2, np.full_like(r2, f32(-1.6666654611e-1)))
        s = fma((ps * r2).astype(np.float32), r, r)
        pc = fma(r2, f32(2.443315711809948e-5), np.full_like(r2, f32(-1.388731625493765e-3)))
        pc = fma(pc, r2, np.full_like(r2, f32(4.166664568298827e-2)))
        c = fma((pc * r2).astype(np.float32), r2, fma(r2, f32(-0.5), np.ones_like(r2)))
        swap = (q & 1) == 1
        ss, cc = np.where(swap, c, s), np.where(swap, s, c)
        return np.where((q & 2) == 2, -ss, ss).astype(np.float32), np.where(((q + 1) & 2) == 2, -cc, cc).astype(np.float32)

    rng = np.random.default_rng(0)
    for hi in (8.0, 1300.0, 1.0e5):
        x = rng.uniform(-hi, hi, 500000).astype(np.float32)
        s, c = sincos(x)
        assert np.abs(s - np.sin(x.astype(np.float64))).max() < 1.5e-7
        assert np.abs(c - np.cos(x.astype(np.float64))).max() < 1.5e-7
