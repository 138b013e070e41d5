"""Analytic known-answer tests for the CPU oracle (SURVEY.md section 8c list) and the cross-check of
its two independent restatements (C + OpenMP vs vectorised PyTorch-CPU)."""
import math

import numpy as np
import pytest
import torch

from autovfx_amd import scenes
from autovfx_amd.cameras import Camera
from autovfx_amd.scenes import GaussianCloud
from oracle import cpu_oracle, torch_splat

from helpers import oracle_kwargs


def one_gaussian(pos, scale, opacity, color):
    return GaussianCloud(torch.tensor([pos], dtype=torch.float32), torch.tensor([[opacity]]),
                         torch.full((1, 3), float(scale)), torch.tensor([[1.0, 0.0, 0.0, 0.0]]), None,
                         torch.tensor([color], dtype=torch.float32), 0)


def test_isotropic_gaussian_alpha_profile():
    W = H = 65
    cam = scenes.c1_camera(W, H)
    s, o = 0.05, 0.8
    out = cpu_oracle.forward(**oracle_kwargs(one_gaussian([0, 0, 0], s, o, [1.0, 0.5, 0.25]), cam))
    fx = W / (2 * cam.tanfovx)
    var = (s * fx / 4.0) ** 2 + 0.3                       # EWA variance + 0.3 dilation
    c = ((0.0 + 1.0) * W - 1.0) * 0.5
    yy, xx = np.mgrid[0:H, 0:W]
    a = np.minimum(0.99, o * np.exp(-((xx - c) ** 2 + (yy - c) ** 2) / (2 * var)))
    a[a < 1 / 255.0] = 0
    # isotropic => mid^2 - det == 0, so lambda = var + sqrt(0.1) (the max(0.1, .) floor, forward.cu:230)
    assert out["radii"][0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    np.testing.assert_allclose(out["alpha"][0], a, atol=2e-6)
    np.testing.assert_allclose(out["color"][1], 0.5 * a, atol=2e-6)
    np.testing.assert_allclose(out["depth"][0], 4.0 * a, atol=1e-5)
    assert out["num_rendered"] == 4                       # centre 32 +- 4 px straddles two tiles per axis


def test_degree0_sh_colour_rule():
    c = scenes.config_c1(P=200, seed=4)
    out = cpu_oracle.forward(intermediates=True, **oracle_kwargs(c, scenes.c1_camera(64, 64), sh_degree=0))
    vis = out["radii"] > 0
    expect = np.maximum(0.0, np.float32(0.28209479177387814) * c.shs[:, 0].numpy() + np.float32(0.5))
    np.testing.assert_array_equal(out["rgb"][vis], expect[vis].astype(np.float32))


def test_near_plane_cull_is_strict():
    fov = math.radians(60.0)
    cam = Camera.from_Rt(np.eye(3), np.zeros(3), fov, fov, 32, 32)
    z_keep = float(np.nextafter(np.float32(0.2), np.float32(1.0)))
    c = scenes.config_c1(P=2, seed=1)
    c.means3D[:] = torch.tensor([[0.0, 0.0, 0.2], [0.0, 0.0, z_keep]])
    out = cpu_oracle.forward(**oracle_kwargs(c, cam))
    assert list(out["radii"] > 0) == [False, True]
    np.testing.assert_array_equal(cpu_oracle.mark_visible(c.means3D, cam.world_view_transform, cam.full_proj_transform),
                                  [False, True])


def test_front_to_back_and_tie_order():
    cam = scenes.c1_camera(48, 48)

    def two(zs, cols):
        return GaussianCloud(torch.tensor([[0.0, 0.0, z] for z in zs]), torch.full((2, 1), 0.6),
                             torch.full((2, 3), 0.2), torch.tensor([[1.0, 0, 0, 0]] * 2), None, torch.tensor(cols), 0)
    red, green = [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]
    px = lambda cl: cpu_oracle.forward(**oracle_kwargs(cl, cam))["color"][:, 24, 24]
    assert px(two([-1.0, 0.0], [red, green]))[0] > px(two([-1.0, 0.0], [red, green]))[1]
    assert px(two([0.0, -1.0], [red, green]))[1] > px(two([0.0, -1.0], [red, green]))[0]
    assert px(two([0.0, 0.0], [red, green]))[0] > px(two([0.0, 0.0], [red, green]))[1]   # tie: lower index first
    assert px(two([0.0, 0.0], [green, red]))[1] > px(two([0.0, 0.0], [green, red]))[0]


def test_empty_inputs():
    cam = scenes.c1_camera(20, 12)
    empty = GaussianCloud(torch.zeros(0, 3), torch.zeros(0, 1), torch.zeros(0, 3), torch.zeros(0, 4),
                          torch.zeros(0, 16, 3), None, 3)
    out = cpu_oracle.forward(**oracle_kwargs(empty, cam, bg=(0.5, 0.5, 0.5)))
    assert out["num_rendered"] == 0 and not out["color"].any() and out["radii"].shape == (0,)


def test_key_bits_is_the_reference_rule():
    # getHigherMsb (rasterizer_impl.cu:35-50): smallest b with n >> b == 0, found by bisection
    for n in (1, 2, 3, 255, 256, 257, 2040, 8160, 65535, 65536, 2 ** 31 - 1):
        assert cpu_oracle.key_bits(n) == max(1, int(n).bit_length()) or n == 1


def test_invariants_on_c1():
    c, cam = scenes.config_c1(), scenes.c1_camera()
    g = torch.Generator().manual_seed(5)
    c.means3D[:, 2] = torch.linspace(-1.0, 1.0, c.P)[torch.randperm(c.P, generator=g)]   # distinct depths
    out = cpu_oracle.forward(intermediates=True, **oracle_kwargs(c, cam))
    assert out["alpha"].min() >= 0 and out["alpha"].max() < 1
    assert out["num_rendered"] == int(out["tiles_touched"].sum()) == int(out["point_offsets"][-1])
    np.testing.assert_array_equal(out["radii"] > 0, out["tiles_touched"] > 0)
    keys = out["point_list_keys"]
    assert (np.diff(keys.astype(np.int64)) >= 0).all()
    # background linearity: color(bg=1) - color(bg=0) == 1 - alpha
    one = cpu_oracle.forward(**oracle_kwargs(c, cam, bg=(1, 1, 1)))
    np.testing.assert_allclose(one["color"] - out["color"], np.broadcast_to(1 - out["alpha"], (3, 256, 256)), atol=2e-6)
    # permutation of the Gaussians (distinct depths) leaves the image unchanged
    perm = torch.randperm(c.P, generator=torch.Generator().manual_seed(1))
    cp = GaussianCloud(c.means3D[perm], c.opacities[perm], c.scales[perm], c.rotations[perm], c.shs[perm], None, 3)
    shuf = cpu_oracle.forward(**oracle_kwargs(cp, cam))
    assert np.unique(out["depths"]).size == c.P
    np.testing.assert_array_equal(shuf["color"], out["color"])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-6), (torch.float64, None)])
def test_c_oracle_vs_torch_splat(dtype, tol):
    c, cam = scenes.config_c1(P=4000, seed=9), scenes.c1_camera(160, 120)
    kw = oracle_kwargs(c, cam, bg=(0.2, 0.4, 0.6))
    a = cpu_oracle.forward(intermediates=True, **kw)
    b = torch_splat.forward(dtype=dtype, **kw)
    assert a["num_rendered"] == b["num_rendered"]
    np.testing.assert_array_equal(a["radii"], b["radii"].numpy())
    np.testing.assert_array_equal(a["point_list"], b["point_list"].numpy())
    err = np.abs(a["color"] - b["color"].numpy()).max(axis=0)
    if tol is not None:
        assert err.max() <= tol
    else:  # fp64 arithmetic flips a few threshold decisions; everything else agrees to fp32 precision
        assert (err > 1e-4).mean() < 1e-3 and np.median(err) < 1e-6
