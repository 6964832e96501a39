"""GPU (-m gpu): the editing consumers of the field against the REFERENCE's own classes / functions:
tests/golden/texture_edit_v3000.npz = editing/texture_neumesh/texture_neumesh.py:TextureEditableNeuMesh run on reference models,
tests/golden/deform_v3000.npz = editing/render_geometry_editing.py:deform_model (oracle/gen_golden.py `edit` / `deform`)."""
import numpy as np
import pytest

import common
from neumesh_amd import synthetic

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("name,n_ref,rotated", [("r1", 1, False), ("r2", 2, False), ("r2T", 2, True)])
def test_texture_editable_neumesh_matches_reference_fixture(cuda_device, name, n_ref, rotated):
    """neumesh_amd.editing.TextureEditableNeuMesh -- point-wise forward (fused HIP field call + blend) and the frame rendered INSIDE
    nm_render_rays (nm_edit.h) and by the staged renderer -- against what the reference's class returned on the same scene: one / two
    texture references with overlapping painted regions, with and without rigid transforms.  Tolerances: the field's (5e-6, plus the
    fp32 sensitivity of the 2^7-band ds embedding for far points) point-wise, north_star's 1e-4 for the rendered frame."""
    import torch
    from neumesh_amd.renderer import fusable_edit_model, make_render_cfg, render_rays_staged, volume_render
    f, fx, rf = common.golden("texture_edit_v3000"), common.golden("field_v3000"), common.golden("render_v3000_dtu")
    mesh = common.scene_mesh(int(f["V"]))
    wrap, main = common.edit_model(mesh, common.scene_state(mesh), n_ref, rotated, cuda_device)
    q, dirs = _t(fx["q"], cuda_device), _t(fx["dirs"], cuda_device)
    with torch.no_grad():
        sdf, rgb = wrap(q, dirs)
    np.testing.assert_allclose(sdf.cpu().numpy(), f[name + ".sdf"], atol=5e-6)
    err = np.abs(rgb.cpu().numpy() - f[name + ".rgb"])
    assert np.all(err <= 5e-6 + 1e-4 * np.abs(fx["ds"])), float(err.max())
    painted = f[name + ".painted"]
    assert float(np.abs(rgb.cpu().numpy() - fx["rgb"])[painted].max()) > 0.1
    # with autograd on (the differentiable forms of the models' methods) the same numbers come out
    sdf_g, rgb_g = wrap(q.clone().requires_grad_(True), dirs)
    assert np.all(np.abs(rgb_g.detach().cpu().numpy() - f[name + ".rgb"]) <= 2e-5 + 1e-4 * np.abs(fx["ds"]))
    ro, rd = _t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device)
    kw = dict(calc_normal=True, perturb=False, detailed_output=False, N_samples=64, N_importance=64)
    assert fusable_edit_model(wrap)
    with torch.no_grad():
        img, depth, ex = volume_render(ro, rd, wrap, rayschunk=4096, **kw)                       # blend inside nm_render_rays
        st = render_rays_staged(wrap, ro, rd, make_render_cfg(calc_normal=True), 4096, 1 << 20)  # wrapper's forward() per stage
    for got, label in ((img, "fused"), (st["rgb"], "staged")):
        np.testing.assert_allclose(got.cpu().numpy(), f[name + ".rgb_render"], atol=1e-4, err_msg=label)
    np.testing.assert_allclose(depth.cpu().numpy(), f[name + ".depth"], atol=1e-4)
    np.testing.assert_allclose(ex["mask_volume"].cpu().numpy(), f[name + ".acc"], atol=1e-4)
    np.testing.assert_allclose(ex["normals_volume"].cpu().numpy(), f[name + ".normals"], atol=1e-4)
    assert float(np.abs(img.cpu().numpy() - f["main.rgb_render"]).max()) > 0.1                   # an edited frame, not the main model's


@pytest.mark.parametrize("fix_indicator", [False, True])
def test_deform_model_matches_reference_fixture(cuda_device, fix_indicator):
    """neumesh_amd.editing.deform_model (same arguments as editing/render_geometry_editing.py:37-67): mesh index rebuilt on the
    deformed mesh (device octree build), indicator vectors rotated with each vertex normal -- by angle * sin(angle), the reference's
    un-normalised rotation vector --, negated where the normal flips exactly, untouched where it does not move; then the field and a
    rendered frame on the deformed model against what the reference produced after ITS deform_model."""
    import torch
    from neumesh_amd.editing import deform_model
    from neumesh_amd.renderer import volume_render
    f, fx, rf = common.golden("deform_v3000"), common.golden("field_v3000"), common.golden("render_v3000_dtu")
    base, dmesh, snap = synthetic.deformed_blob(common.scene_mesh(int(f["V"])))
    assert np.array_equal(dmesh.vertices, f["deformed_vertices"]) and np.array_equal(dmesh.vertex_normals, f["deformed_normals"])
    state = common.scene_state(base)
    model = common.make_model(base, state, cuda_device)
    name = "fix" if fix_indicator else "rot"
    deform_model(common.MeshObj(dmesh), model, cuda_device, fix_indicator=fix_indicator)
    assert isinstance(model.indicator_vector, torch.nn.Parameter)
    ind = model.indicator_vector.detach().cpu().numpy()
    np.testing.assert_allclose(ind, f[name + ".indicator"], atol=2e-6)
    if not fix_indicator:
        assert np.array_equal(ind[snap[:3]], -state["indicator_vector"][snap[:3]])       # cos == -1: negated
        assert np.array_equal(ind[snap[3:]], state["indicator_vector"][snap[3:]])        # rotation vector 0: identity
        assert np.median(np.linalg.norm(ind - state["indicator_vector"], axis=-1)) > 0.03
    q = _t(fx["q"], cuda_device)
    with torch.no_grad():
        ds, idx, _ = model.compute_distance(q)
        sdf = model.forward_density_only(q)
        img, depth, ex = volume_render(_t(rf["rays_o"], cuda_device), _t(rf["rays_d"], cuda_device), model, calc_normal=True, perturb=False,
                                       detailed_output=False, N_samples=64, N_importance=64, rayschunk=4096)
    assert np.array_equal(idx.cpu().numpy(), f[name + ".idx"])                           # K-NN on the rebuilt index: bit-exact
    np.testing.assert_allclose(ds.cpu().numpy(), f[name + ".ds"], atol=3e-6)
    np.testing.assert_allclose(sdf.cpu().numpy(), f[name + ".sdf"], atol=5e-6)
    np.testing.assert_allclose(img.cpu().numpy(), f[name + ".rgb_render"], atol=1e-4)
    np.testing.assert_allclose(depth.cpu().numpy(), f[name + ".depth"], atol=1e-4)
    np.testing.assert_allclose(ex["mask_volume"].cpu().numpy(), f[name + ".acc"], atol=1e-4)
    np.testing.assert_allclose(ex["normals_volume"].cpu().numpy(), f[name + ".normals"], atol=1e-4)


def test_texture_edit_headline_scale_surface_scene_matches_reference_fixture(cuda_device):
    """Row f2 at the scale and on the scene the bench times it on (VERDICT r3 missing #2): tests/golden/texture_edit_v140k_surf.npz = the
    reference's TextureEditableNeuMesh (two references, overlapping caps, rigid transforms) on the surface scene at V = 140 000, its frame
    of the first 384 fixture rays by the reference's SingleRenderer.  Geometry outputs are the main model's (depth / acc / normals equal
    to render_v140k_surf.npz's on the same rays, on both sides).  The sample placement on this scene is sensitive to the last bit (§5 of
    DESIGN.md), so the colour gate is taken on the rays where the product's MAIN-model frame agrees with the reference's (<= 1e-5:
    the same samples) -- there the edited frame must agree to 1e-4 -- plus the statistical gate of the main-model test on all rays."""
    import torch
    from neumesh_amd.renderer import fusable_edit_model, volume_render
    f, base = common.golden("texture_edit_v140k_surf"), common.golden("render_v140k_surf")
    n = int(f["n_rays"])
    mesh = common.scene_mesh(int(f["V"]))
    state = common.surface_state(mesh)
    assert str(f["state_sha256"]) == common.state_digest({k: v for k, v in state.items() if k not in ("geometry_features", "color_features", "indicator_vector")})
    mlp = {k: v for k, v in state.items() if k not in ("geometry_features", "color_features", "indicator_vector")}
    from neumesh_amd.editing import TextureEditableNeuMesh
    masks, feats, T_list = synthetic.edit_scene(mesh.vertices, 2, True)
    assert np.array_equal(masks.sum(1), f["mask_sum"])
    main = common.make_model(mesh, state, cuda_device)
    refs = [common.make_model(mesh, {**state, **synthetic.reference_color_state(mlp, i, gain=1.5)}, cuda_device) for i in range(2)]
    wrap = TextureEditableNeuMesh(main, refs, torch.from_numpy(masks).to(cuda_device), torch.from_numpy(feats).to(cuda_device),
                                  [torch.from_numpy(t).to(cuda_device) for t in T_list]).eval()
    assert fusable_edit_model(wrap) and abs(float(main.forward_s()) - float(f["s"])) <= 1e-3
    ro, rd = _t(base["rays_o"][:n], cuda_device), _t(base["rays_d"][:n], cuda_device)
    kw = dict(calc_normal=True, perturb=False, detailed_output=False, N_samples=64, N_importance=64, rayschunk=65536)
    with torch.no_grad():
        img, depth, ex = volume_render(ro, rd, wrap, **kw)
        img0, depth0, ex0 = volume_render(ro, rd, main, **kw)
    assert torch.equal(depth, depth0) and torch.equal(ex["mask_volume"], ex0["mask_volume"]) and torch.equal(ex["normals_volume"], ex0["normals_volume"])
    e_main = np.abs(img0.cpu().numpy() - base["rgb"][:n]).max(-1)
    e_edit = np.abs(img.cpu().numpy() - f["rgb"]).max(-1)
    same = e_main <= 1e-5
    moved = np.abs(f["rgb"] - base["rgb"][:n]).max(-1) > 1e-2
    print(f"edited frame at V = 140 000: {int(same.sum())}/{n} rays with the reference's samples, max error among them {e_edit[same].max():.2e}; "
          f"all rays: median {np.median(e_edit):.1e}, beyond 1e-4: {int((e_edit > 1e-4).sum())} (main-model frame: {int((e_main > 1e-4).sum())}); "
          f"rays the edit moves by > 1e-2: {int(moved.sum())}, of them with the reference's samples: {int((moved & same).sum())}")
    assert same.sum() >= 0.9 * n and (moved & same).sum() >= 50
    assert np.median(e_edit) <= 1e-6 and (e_edit > 1e-4).mean() <= (base["self_err_1ulp"][:n] > 1e-4).mean() + 0.02
    # ... and without the sampler: the edited field + compositing on the REFERENCE's own sample depths of these rays (the edit does not move a
    # sample: geometry is the main model's), through the wrapper's forward -- every ray within 1e-4.  (Agreement of the main-model pixel is not
    # enough to call the samples equal: the painted share of a point jumps where its 8-neighbour set changes, so the edited colour reacts to a
    # last-bit move of a sample that the main colour does not notice -- 3 such rays here, up to 2.8e-4.)
    from neumesh_amd.renderer import make_render_cfg, render_at_depths
    with torch.no_grad():
        tail = render_at_depths(wrap, ro, rd, _t(base["d_all"][:n], cuda_device), make_render_cfg(calc_normal=True))
    e_tail = np.abs(tail["rgb"].cpu().numpy() - f["rgb"]).max(-1)
    print(f"  on the reference's own depths: max colour error {e_tail.max():.2e}, depth {np.abs(tail['depth_volume'].cpu().numpy() - f['depth_volume']).max():.2e}")
    assert e_tail.max() <= 1e-4 and float(e_tail[moved].max()) <= 1e-4
    np.testing.assert_allclose(tail["mask_volume"].cpu().numpy(), f["mask_volume"], atol=1e-4)
    np.testing.assert_allclose(tail["normals_volume"].cpu().numpy(), f["normals_volume"], atol=1e-4)
