"""The fp64 gradient truth (oracle/gsr_oracle.c: gsro_backward_f64) pinned from two sides:

* against float64 AUTOGRAD of an independently written dense splat (no tiles, no lists, every Gaussian against every pixel,
  cumprod transmittance) on a small scene built so that tiling, the near plane and saturation do not act -- the truth then is
  simply d(loss)/d(parameters), and PyTorch differentiates a different program to get it;
* against the reference's own fp32 backward (the CPU oracle = backward.cu bit for bit, and the committed golden vectors) on
  well-conditioned scenes, where fp32 is within 3e-4 of it.
"""
import math
import os

import numpy as np
import pytest
import torch

from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
from autovfx_amd.scenes import GaussianCloud
from oracle import cpu_oracle, torch_splat

from helpers import gradient_errors, oracle_kwargs
from test_oracle_backward import GOLDEN_BW, load_bw_case, pixel_grads


def dense_splat(means3D, opac, scales, rots, shs, cam, bg, deg):
    """float64, differentiable, O(P * pixels): SURVEY.md appendix A.3 without tiles."""
    dt = torch.float64
    V, PM = cam.world_view_transform.to(dt), cam.full_proj_transform.to(dt)
    campos = cam.camera_center.to(dt)
    W, H = cam.image_width, cam.image_height
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat((means3D, ones), 1) @ PM
    pv = (torch.cat((means3D, ones), 1) @ V)[:, :3]
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    r, x, y, z = rots.unbind(1)
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), 1).view(-1, 3, 3)
    L = R @ torch.diag_embed(scales)
    Sig = L @ L.transpose(1, 2)
    fx, fy = W / (2.0 * cam.tanfovx), H / (2.0 * cam.tanfovy)
    tz = pv[:, 2]
    J = torch.zeros(P, 2, 3, dtype=dt)
    J[:, 0, 0] = fx / tz
    J[:, 0, 2] = -fx * pv[:, 0] / (tz * tz)
    J[:, 1, 1] = fy / tz
    J[:, 1, 2] = -fy * pv[:, 1] / (tz * tz)
    JW = J @ V[:3, :3].t()
    cov = JW @ Sig @ JW.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    cxx, cxy, cyy = c / det, -b / det, a / det
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    d = means3D - campos[None]
    rgb = torch_splat._sh_rgb(deg, d / d.norm(dim=1, keepdim=True), shs)
    order = torch.argsort(tz.detach())
    gy, gx = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    dx = px[order, None, None] - gx[None]
    dy = py[order, None, None] - gy[None]
    power = -0.5 * (cxx[order, None, None] * dx * dx + cyy[order, None, None] * dy * dy) - cxy[order, None, None] * dx * dy
    alpha = torch.clamp_max(opac.reshape(-1)[order, None, None] * torch.exp(power), 0.99)
    alpha = torch.where((power > 0) | (alpha < 1.0 / 255.0), torch.zeros((), dtype=dt), alpha)
    Tin = torch.cumprod(torch.cat((torch.ones(1, H, W, dtype=dt), 1 - alpha), 0), 0)
    w = alpha * Tin[:-1]
    color = torch.einsum("phw,pc->chw", w, rgb[order]) + Tin[-1][None] * bg.to(dt)[:, None, None]
    depth = (w * tz[order, None, None]).sum(0, keepdim=True)
    return color, depth, (1 - Tin[-1])[None], float(Tin[-1].detach().min())


def small_smooth_scene(seed):
    g = torch.Generator().manual_seed(seed)
    P = 40
    cam = scenes.c1_camera(48, 32)   # at (0, 0, -4) looking down +z
    means = (torch.rand(P, 3, generator=g) - 0.5) * torch.tensor([2.4, 1.6, 1.0])
    scales = torch.exp(torch.randn(P, 3, generator=g) * 0.3 + math.log(0.12))
    rots = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=1)
    opac = torch.rand(P, 1, generator=g) * 0.25 + 0.05     # <= 0.3: below 1/255 outside the 3 sigma rectangle (0.3 exp(-4.5) < 1/255)
    shs = torch.randn(P, 16, 3, generator=g) * torch.tensor([1.0] + [0.2] * 15)[None, :, None]
    return GaussianCloud(means, opac, scales, rots, shs, None, 3), cam


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_truth_equals_float64_autograd_of_a_dense_splat(seed):
    cloud, cam = small_smooth_scene(seed)
    pg = pixel_grads(cam, seed)
    bg = torch.tensor([0.3, 0.1, 0.2])
    leaves = [t.to(torch.float64).requires_grad_(True) for t in (cloud.means3D, cloud.opacities, cloud.scales, cloud.rotations, cloud.shs)]
    color, depth, alpha, t_min = dense_splat(*leaves, cam, bg, 3)
    assert t_min > 1e-3, "a pixel came close to saturation: the dense splat does not model the early stop"
    kw = oracle_kwargs(cloud, cam, bg=bg.numpy())
    fwd = cpu_oracle.forward(**kw)
    assert int((fwd["radii"] > 0).sum()) == cloud.P
    assert np.abs(fwd["color"] - color.detach().numpy()).max() < 2e-5     # the two programs render the same image
    assert np.abs(fwd["alpha"] - alpha.detach().numpy()).max() < 2e-5
    loss = ((color * torch.from_numpy(pg["dL_dcolor"])).sum() + (depth * torch.from_numpy(pg["dL_ddepth"])).sum()
            + (alpha * torch.from_numpy(pg["dL_dalpha"])).sum())
    loss.backward()
    kw.update(pg)
    truth = cpu_oracle.backward_f64(**kw)
    for name, leaf in zip(("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh"), leaves):
        want, got = leaf.grad.numpy().reshape(-1), truth[name].reshape(-1)
        scale = np.abs(want).max()
        # (the truth stands on the fp32 forward state -- means2D, conic, colours rounded to fp32 -- hence 1e-5, not 1e-12)
        assert np.abs(got - want).max() <= 2e-5 * scale, (name, np.abs(got - want).max() / scale)


def test_truth_is_near_the_reference_fp32_backward_on_well_conditioned_scenes():
    for cloud, cam, okw in ((scenes.config_c1(P=4000, seed=3), scenes.c1_camera(128, 96), dict(bg=(0.1, 0.2, 0.3))),
                            (scenes.config_c1(P=1500, seed=6), orbit_cameras(8, 96, 64)[5], dict(scale_modifier=1.6, sh_degree=2))):
        kw = oracle_kwargs(cloud, cam, **okw)
        kw.update(pixel_grads(cam, 2))
        ref, truth = cpu_oracle.backward(**kw), cpu_oracle.backward_f64(**kw)
        for k in (k for k in truth if k.startswith("dL_")):
            e_ref, _, _, scale = gradient_errors(ref[k], ref[k], truth[k])
            assert e_ref <= 3e-4 * scale + 1e-9, (k, e_ref / max(scale, 1e-30))


@pytest.mark.parametrize("path", GOLDEN_BW, ids=[os.path.basename(p)[:-4] for p in GOLDEN_BW])
def test_truth_is_near_the_reference_golden_vectors(path):
    kw, ref = load_bw_case(path)
    truth = cpu_oracle.backward_f64(**kw)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dsh", "dL_dmeans3D", "dL_dconic", "dL_ddepths"):
        e_ref, _, _, scale = gradient_errors(ref[k], ref[k], truth[k])
        assert e_ref <= 3e-4 * scale + 1e-9, (k, e_ref / max(scale, 1e-30))
    for k in ("dL_dscales", "dL_drotations", "dL_dcov3D"):   # through 1 / (denom^2 + 1e-7): flat SuGaR-style cases are looser
        e_ref, _, _, scale = gradient_errors(ref[k], ref[k], truth[k])
        assert e_ref <= 5e-3 * scale + 1e-9, (k, e_ref / max(scale, 1e-30))


def test_truth_of_an_empty_and_of_an_all_culled_scene_is_zero():
    cam = scenes.c1_camera(32, 32)
    cloud = scenes.config_c1(P=50, seed=1)
    cloud.means3D[:, 2] = -10.0      # behind the camera at z = -4
    kw = oracle_kwargs(cloud, cam)
    kw.update(pixel_grads(cam, 1))
    truth = cpu_oracle.backward_f64(**kw)
    assert all(not np.any(v) for v in truth.values())
    assert not any(np.any(v) for v in cpu_oracle.fp32_noise(truth, **kw).values())


def test_noise_yardstick_is_small_where_fp32_is_good_and_large_on_needles():
    """cpu_oracle.fp32_noise: on a well-conditioned scene the yardstick stays below the plain 2e-4 bar (the truth-based bar then
    is the old one); on the flat SuGaR-style Gaussians it says what the reference's own fp32 shows -- errors of 1e-3 of the scale
    in dL_dscales -- and it is a pure function of the scene (same arrays twice)."""
    cloud, cam = scenes.config_c1(P=3000, seed=8), scenes.c1_camera(96, 64)
    kw = oracle_kwargs(cloud, cam)
    kw.update(pixel_grads(cam, 4))
    truth = cpu_oracle.backward_f64(**kw)
    noise = cpu_oracle.fp32_noise(truth, **kw)
    for k, v in noise.items():
        scale = float(np.abs(truth[k]).max())
        assert v.max() <= 2e-4 * scale, (k, v.max() / scale)
    cloud, cam = scenes.config_c4(P=20000, seed=3), orbit_cameras(8, 320, 180)[3]
    kw = oracle_kwargs(cloud, cam, bg=(1.0, 1.0, 1.0))
    kw.update(pixel_grads(cam, 5))
    truth, ref = cpu_oracle.backward_f64(**kw), cpu_oracle.backward(**kw)
    noise = cpu_oracle.fp32_noise(truth, **kw)
    again = cpu_oracle.fp32_noise(truth, **kw)
    assert all(np.array_equal(noise[k], again[k]) for k in noise)
    e_ref, _, _, scale = gradient_errors(ref["dL_dscales"], ref["dL_dscales"], truth["dL_dscales"])
    assert noise["dL_dscales"].max() > 0.25 * e_ref, (noise["dL_dscales"].max() / scale, e_ref / scale)
