"""GPU (-m gpu): replay of what the reference's own driver does around the renderer.

tests/golden/render_py_trace.npz was recorded by RUNNING the reference's `render.render_function` (render.py:99-260, unmodified) in the
build container with a recording stand-in for the renderer (oracle/gen_golden.py trace): the camera poses of its spiral path and the
rays `rend_util.get_rays` made of them, the exact keyword arguments it passes to `render_fn` (build_framework's render_kwargs_test plus
show_progress / detailed_output / rayschunk), the stand-in's return values and every image it then wrote.  Here the same calls go to the
PRODUCT: (1) `neumesh_amd.rays.get_rays` on the recorded poses reproduces the recorded rays, (2) `get_model`'s renderer accepts the
recorded call verbatim and returns what render.py consumes (shapes, keys, dtypes, finite values; the consumer's own expressions run
on it), (3) the product's frame assembly turns the stand-in's outputs into the very bytes render.py wrote."""
import json
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def test_reference_render_py_calls_replayed_on_the_product(cuda_device):
    import torch
    from neumesh_amd import frames
    from neumesh_amd.rays import get_rays
    f = common.golden("render_py_trace")
    H, W, n_views = int(f["H"]), int(f["W"]), int(f["n_views"])
    kw = json.loads(str(f["kwargs_json"]))
    assert kw["detailed_output"] is False and kw["show_progress"] is True and kw["rayschunk"] == 4096 and kw["perturb"] is False   # render.py:176,211-218
    # the product's model + SingleRenderer (what neumesh_amd.framework.get_model returns; its config-file path is tests/test_host.py's subject)
    mesh = common.scene_mesh(int(f["V"]))
    model = common.make_model(mesh, common.scene_state(mesh), cuda_device)
    from neumesh_amd.renderer import SingleRenderer
    renderer = SingleRenderer(model)
    written = {str(n): f[f"written_{j}"] for j, n in enumerate(f["written_names"])}
    rgb_imgs, depth_imgs = [], []
    for i in range(n_views):
        c2w, K = torch.from_numpy(f[f"c2w_{i}"]).to(cuda_device), torch.from_numpy(f[f"intrinsics_{i}"]).to(cuda_device)
        # (1) rays: render.py:202-208
        ro, rd, sel = get_rays(c2w, K, H, W, N_rays=-1)
        assert tuple(ro.shape) == tuple(f[f"rays_o_{i}"].shape) == (1, H * W, 3)
        assert np.abs(ro.cpu().numpy() - f[f"rays_o_{i}"]).max() <= 1e-6 and np.abs(rd.cpu().numpy() - f[f"rays_d_{i}"]).max() <= 2e-6
        # (2) the recorded call, verbatim: render.py:210-218
        with torch.no_grad():
            rgb, depth, extras = renderer(ro, rd, **kw)
        assert tuple(rgb.shape) == (1, H * W, 3) and tuple(depth.shape) == (1, H * W) and "normals_volume" in extras
        d_ = depth.data.cpu().reshape(H, W, 1).numpy()            # render.py:219-233, the consumer's own expressions
        d_ = d_ / d_.max()
        img = rgb.data.cpu().reshape(H, W, 3).numpy()
        nrm = extras["normals_volume"].data.cpu().reshape(H, W, 3).numpy() / 2.0 + 0.5
        for a in (d_, img, nrm):
            assert np.isfinite(a).all()
        assert img.min() >= 0.0 and img.max() <= 1.0 + 1e-5 and d_.max() == 1.0
        # (3) what render.py wrote from the STAND-IN's outputs == the product's frame assembly of the same outputs
        out = frames.assemble_images(torch.from_numpy(f[f"rgb_{i}"]).to(cuda_device), torch.from_numpy(f[f"depth_{i}"]).to(cuda_device),
                                     torch.from_numpy(f[f"normals_{i}"]).to(cuda_device), H=H, W=W, bgr=True)
        assert np.array_equal(out["rgb"].cpu().numpy(), written[f"cv2.imwrite:trace_rgb_{i:03d}.png"])          # render.py:234-242 (BGR)
        assert np.array_equal(out["normal"].cpu().numpy(), written[f"imageio.imwrite:trace_normal_{i:03d}.png"])  # :243-249
        rgb_imgs.append(out["rgb"].cpu().numpy()[..., ::-1])
        depth_imgs.append(out["depth"].cpu().numpy())
    assert np.array_equal(np.stack(rgb_imgs), written[f"imageio.mimwrite:trace_rgb_{H}x{W}_{n_views}_spiral.mp4"])      # render.py:251-263
    assert np.array_equal(np.stack(depth_imgs), written[f"imageio.mimwrite:trace_depth_{H}x{W}_{n_views}_spiral.mp4"])
