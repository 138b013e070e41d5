"""gsr_forward_raw (second half): the forward pass from a model's RAW parameter tensors.

The bar is BIT-EXACT, end to end, against what an unchanged ``render()`` does on this GPU: activate in PyTorch
(``exp`` / ``sigmoid`` / ``F.normalize`` / ``cat`` / ``get_normal``, ``scene/gaussian_model.py:95-128``,
``utils/general_utils.py:78-157``), call the rasterizer twice, post-process in PyTorch
(``gaussian_renderer/__init__.py:118-208``).  ``ReferenceShapedModel`` below restates the reference's ``GaussianModel``
getters line for line (no memo, activation functions kept as attributes by ``setup_functions``).
"""
import math

import pytest
import torch

from autovfx_amd import gaussian_model as gm
from autovfx_amd import renderer, scenes
from autovfx_amd.cameras import orbit_cameras, sugar_orbit_cameras

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class ReferenceShapedModel:
    """``scene/gaussian_model.py:25-128`` of the reference: raw tensors + getters that recompute on every access."""

    def setup_functions(self):
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def __init__(self, sh_degree, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, active=None):
        self.active_sh_degree = sh_degree if active is None else active
        self.max_sh_degree = sh_degree
        self._xyz, self._scaling, self._rotation, self._opacity = xyz, log_scales, rotations, opacity_logits
        self._features_dc, self._features_rest = features_dc, features_rest
        self.setup_functions()

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_normal(self, dir_pp_normalized=None):
        normal_axis = self.get_minimum_axis
        normal_axis, _ = gm.flip_align_view(normal_axis, dir_pp_normalized)
        return normal_axis / normal_axis.norm(dim=1, keepdim=True)

    @property
    def get_minimum_axis(self):
        return gm.get_minimum_axis(self.get_scaling, self.get_rotation)


def raw_model(P, seed, sh_degree=3, M=16, nasty=True, cloud=None):
    """Raw parameters as a trained model holds them: un-normalised quaternions, log scales, opacity logits; with ``nasty``
    also the corner cases of the activations (tied scales in every pattern, tiny / huge quaternion norms, saturated
    logits)."""
    c = cloud if cloud is not None else scenes.config_c1(P=P, seed=seed)
    g = torch.Generator().manual_seed(1000 + seed)
    P = c.means3D.shape[0]
    ls = torch.log(c.scales)
    rot = c.rotations * (0.5 + torch.rand(P, 1, generator=g) * 3.0) + 0.01 * torch.randn(P, 4, generator=g)
    op = gm.inverse_sigmoid(c.opacities.reshape(P, 1).clamp(1e-6, 1 - 1e-6))
    if nasty:
        ls[2::17, 1] = ls[2::17, 0]                              # s0 == s1
        ls[4::19, 2] = ls[4::19, 0]                              # s0 == s2
        ls[6::23, 1] = ls[6::23, 0]; ls[6::23, 2] = ls[6::23, 0]  # all equal
        ls[8::29, 2] = ls[8::29, 1]                              # s1 == s2
        rot[1::31] *= 1e-4
        rot[3::37] *= 300.0
        op[5::41] = 30.0
        op[7::43] = -30.0
        op[9::47] = 95.0
    dc = c.shs[:, :1].clone().contiguous()
    rest = c.shs[:, 1:M].clone().contiguous()
    t = lambda a: a.to(DEV).contiguous()
    return ReferenceShapedModel(sh_degree, t(c.means3D), t(ls), t(rot), t(op), t(dc), t(rest))


def torch_activated(m, cam):
    """What the reference's render() hands to its two rasterizer passes (gaussian_renderer/__init__.py:118-171)."""
    xyz = m.get_xyz
    dir_pp = xyz - cam.camera_center.repeat(m.get_features.shape[0], 1)
    dirn = dir_pp / dir_pp.norm(dim=1, keepdim=True)
    normals = m.get_normal(dir_pp_normalized=dirn) * 0.5 + 0.5
    return dict(means3D=xyz, opacity=m.get_opacity, scales=m.get_scaling, rotations=m.get_rotation, shs=m.get_features,
                normals=normals.contiguous())


def call_activated(m, cam, bg, inference, scale_modifier=1.0):
    from diff_gaussian_rasterization import _C
    a = torch_activated(m, cam)
    absent = torch.Tensor([])
    return _C.rasterize_gaussians_extra(bg, a["means3D"], absent, a["opacity"], a["scales"], a["rotations"], scale_modifier, absent,
                                        cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                                        cam.image_height, cam.image_width, a["shs"], m.active_sh_degree, cam.camera_center,
                                        False, False, a["normals"], inference=inference), a


def call_raw(m, cam, bg, inference, scale_modifier=1.0, want_normal=True):
    from diff_gaussian_rasterization import _C
    return _C.rasterize_gaussians_raw(bg, m._xyz, m._scaling, m._rotation, m._opacity, m._features_dc, m._features_rest,
                                      scale_modifier, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                                      cam.image_height, cam.image_width, m.active_sh_degree, cam.camera_center, False, False,
                                      want_normal=want_normal, inference=inference)


def bits_equal(a, b):
    return torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


def test_device_activations_are_pytorchs_bit_for_bit():
    """exp, sigmoid, F.normalize and the view normal as the raw kernels evaluate them, read back from the geometry arena of
    a full call, against the PyTorch kernels on the same device -- including scale ties, degenerate quaternions and
    saturated logits.  (The per-splat record holds opacity directly; scales and rotations are checked through what they
    produce: conic, radii and the normal.)"""
    from diff_gaussian_rasterization import _C
    from autovfx_amd import _lib
    cam = orbit_cameras(8, 320, 200)[2].to(DEV)
    m = raw_model(60_000, 3)
    bg = torch.tensor([0.2, 0.3, 0.1], device=DEV)
    with torch.no_grad():
        raw = call_raw(m, cam, bg, inference=False)
        lay = _C.last_layout()
        act, a = call_activated(m, cam, bg, inference=False)
        lay_act = _C.last_layout()
    torch.cuda.synchronize()
    P = m._xyz.shape[0]
    assert raw[0] == act[0]
    radii = raw[4]
    assert torch.equal(radii, act[4])
    geom = raw[5]
    off = lay["geom"]
    rec = geom[off["raster"]: off["raster"] + 32 * P].view(torch.float32).view(P, 8)
    off_a = lay_act["geom"]
    rec_a = act[5][off_a["raster"]: off_a["raster"] + 32 * P].view(torch.float32).view(P, 8)
    vis = radii > 0
    assert int(vis.sum()) > P // 2
    assert bits_equal(rec[vis], rec_a[vis]), "per-splat raster records differ between raw and PyTorch-activated inputs"
    assert bits_equal(rec[vis][:, 5], a["opacity"].reshape(-1)[vis]), "in-kernel sigmoid differs from torch.sigmoid"
    assert off["view_normals"] != 0
    vn = geom[off["view_normals"]: off["view_normals"] + 12 * P].view(torch.float32).view(P, 3)
    # written for every splat with a tile rectangle; all of those have a radius
    bins = geom[off["splat_bins"]: off["splat_bins"] + 16 * P].view(torch.int32).view(P, 4)
    emits = bins[:, 1] != 0
    assert int(emits.sum()) > P // 2 and bool((vis | ~emits).all())
    assert bits_equal(vn[emits], a["normals"][emits]), "in-kernel view normal differs from get_normal(dir) * 0.5 + 0.5"
    # the tie patterns were present among the checked normals
    s = a["scales"][emits]
    assert int(((s[:, 0] == s[:, 1]) & (s[:, 0] == s[:, 2])).sum()) > 100 and int((s[:, 0] == s[:, 1]).sum()) > 200
    for i in (1, 2, 3, 8):
        assert bits_equal(raw[i], act[i]), f"output {i} differs"


@pytest.mark.parametrize("degree,M", [(0, 16), (1, 16), (2, 16), (3, 16), (3, 9), (0, 1), (2, 4)])
@pytest.mark.parametrize("inference", [False, True])
def test_raw_call_equals_activated_call(degree, M, inference):
    cam = orbit_cameras(10, 208, 120)[(degree + M) % 10].to(DEV)
    m = raw_model(20_000, 10 + degree, sh_degree=degree, M=M)
    bg = torch.tensor([0.0, 0.4, 0.9], device=DEV)
    with torch.no_grad():
        raw = call_raw(m, cam, bg, inference)
        act, _ = call_activated(m, cam, bg, inference)
    torch.cuda.synchronize()
    assert raw[0] == act[0]
    for i in (1, 2, 3, 4, 8):
        assert bits_equal(raw[i], act[i]), f"output {i} differs (degree {degree}, M {M}, inference {inference})"


def test_raw_without_normal_and_with_scale_modifier():
    cam = orbit_cameras(10, 160, 96)[7].to(DEV)
    m = raw_model(8_000, 21)
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        raw = call_raw(m, cam, bg, True, scale_modifier=1.7, want_normal=False)
        act, _ = call_activated(m, cam, bg, True, scale_modifier=1.7)
    assert raw[8] is None
    for i in (1, 2, 3, 4):
        assert bits_equal(raw[i], act[i])


def test_raw_argument_errors():
    from diff_gaussian_rasterization import _C
    cam = orbit_cameras(4, 64, 48)[0].to(DEV)
    m = raw_model(100, 1, nasty=False)
    bg = torch.zeros(3, device=DEV)
    args = lambda **kw: dict(dict(xyz=m._xyz, ls=m._scaling, rot=m._rotation, op=m._opacity, dc=m._features_dc, rest=m._features_rest), **kw)

    def go(a):
        return _C.rasterize_gaussians_raw(bg, a["xyz"], a["ls"], a["rot"], a["op"], a["dc"], a["rest"], 1.0, cam.world_view_transform,
                                          cam.full_proj_transform, cam.tanfovx, cam.tanfovy, 48, 64, 3, cam.camera_center, False, False)
    with pytest.raises(RuntimeError, match="features_dc"):
        go(args(dc=m._features_dc.reshape(-1, 3)))
    with pytest.raises(RuntimeError, match="log_scales"):
        go(args(ls=m._scaling[:50]))
    with pytest.raises(RuntimeError, match="float32"):
        go(args(rot=m._rotation.double()))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        go(args(xyz=m._xyz.cpu()))
    empty = _C.rasterize_gaussians_raw(bg, m._xyz[:0], m._scaling[:0], m._rotation[:0], m._opacity[:0], m._features_dc[:0],
                                       m._features_rest[:0], 1.0, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx,
                                       cam.tanfovy, 48, 64, 3, cam.camera_center, False, False)
    assert empty[0] == 0 and float(empty[1].abs().max()) == 0.0 and empty[8].shape == (3, 48, 64)   # P == 0: zeros, not background


def reference_shaped_render(cam, m, bg):
    """render() the reference's way: PyTorch activations and post-processing, two rasterizer calls."""
    saved = renderer.FUSE_ELEMENTWISE
    renderer.FUSE_ELEMENTWISE = False
    try:
        with torch.no_grad():
            return renderer.render(cam, m, renderer.PipelineParams, bg)
    finally:
        renderer.FUSE_ELEMENTWISE = saved


RENDER_KEYS = ("render", "depth", "normal", "pseudo_normal", "radii", "visibility_filter")


@pytest.mark.parametrize("size,P", [((320, 200), 40_000), ((97, 61), 9_000), ((640, 360), 150_000)])
def test_render_from_raw_parameters_is_the_reference_shaped_render_bit_for_bit(size, P):
    """The whole dictionary render() returns: RGBA, depth, unit normal map, pseudo normals from the depth map, radii,
    visibility -- fused raw path (one rasterizer call, one elementwise kernel) against the reference's structure run through
    PyTorch on the same GPU.  ``torch.equal`` on all of them."""
    cam = orbit_cameras(12, *size)[5].to(DEV)
    m = raw_model(P, 30 + size[0])
    assert renderer.raw_parameters(m) is not None
    bg = torch.tensor([0.3, 0.1, 0.2], device=DEV)
    with torch.no_grad():
        got = renderer.render(cam, m, renderer.PipelineParams, bg)
    want = reference_shaped_render(cam, m, bg)
    torch.cuda.synchronize()
    for k in RENDER_KEYS:
        assert got[k].shape == want[k].shape and got[k].dtype == want[k].dtype, k
        same = bits_equal(got[k], want[k]) if got[k].dtype == torch.float32 else torch.equal(got[k], want[k])
        if not same:
            d = (got[k].float() - want[k].float()).abs()
            raise AssertionError(f"{k}: {int((d > 0).sum())} of {d.numel()} elements differ, max {float(d.max()):.3e}")


@pytest.mark.parametrize("seed", range(8))
def test_wild_raw_parameters_render_like_the_pytorch_path(seed):
    """Raw tensors no optimiser would leave behind -- log-scales from -16 to 9 (exp from 1e-7 to 8 000), logits from -100 to 100
    (sigmoid exactly 0 and exactly 1), quaternions with norms from 1e-14 (below F.normalize's eps) to 1e3 and exact zeros, SH
    coefficients up to +-50, a sprinkle of NaN and inf in each -- through the fused raw path and through PyTorch's activations and
    post-processing: the same bits wherever the result is a number, NaN where it is NaN (payloads aside)."""
    g = torch.Generator().manual_seed(4000 + seed)
    P = 6000
    r = lambda *shape: torch.rand(*shape, generator=g)
    n = lambda *shape: torch.randn(*shape, generator=g)
    xyz = n(P, 3) * torch.tensor([2.0, 2.0, 2.0])
    ls = r(P, 3) * 25 - 16
    rot = n(P, 4) * 10 ** (r(P, 1) * 17 - 14)
    rot[r(P) < 0.02] = 0.0
    op = r(P, 1) * 200 - 100
    dc, rest = n(P, 1, 3) * 10 ** (r(P, 1, 1) * 3 - 1.3), n(P, 15, 3) * 10 ** (r(P, 1, 1) * 3 - 1.3)
    if seed >= 4:   # ... and non-finite entries, 0.5 % of the Gaussians in one array each
        for t in (xyz, ls, rot, op, dc, rest):
            rows = torch.randperm(P, generator=g)[:5]
            flat = t.reshape(P, -1)
            flat[rows, torch.randint(0, flat.shape[1], (5,), generator=g)] = torch.tensor([float("nan"), float("inf"), -float("inf"), float("nan"), float("inf")])
    m = ReferenceShapedModel(3, *(t.to(DEV).contiguous() for t in (xyz, ls, rot, op, dc, rest)))
    assert renderer.raw_parameters(m) is not None
    cam = orbit_cameras(12, 233, 141)[seed].to(DEV)
    bg = torch.tensor([0.3, 0.1, 0.2], device=DEV)
    with torch.no_grad():
        got = renderer.render(cam, m, renderer.PipelineParams, bg)
    want = reference_shaped_render(cam, m, bg)
    torch.cuda.synchronize()
    for k in RENDER_KEYS:
        a, b = got[k], want[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if a.dtype != torch.float32:
            assert torch.equal(a, b), k
            continue
        nan_a, nan_b = torch.isnan(a), torch.isnan(b)
        assert torch.equal(nan_a, nan_b), f"{k}: NaN in {int((nan_a != nan_b).sum())} different places"
        assert bits_equal(torch.where(nan_a, torch.zeros_like(a), a), torch.where(nan_b, torch.zeros_like(b), b)), \
            f"{k}: {int(((a != b) & ~nan_a).sum())} elements differ"


def test_render_raw_with_the_sugar_camera_and_inverse_on_the_fly():
    """A camera without the precomputed inverse (the reference's GSCamera has none): c2w comes from
    ``world_view_transform.inverse()`` on the GPU in both paths; off-centre principal point as SuGaR's cameras have."""
    cam = sugar_orbit_cameras(6, 256, 144)[2].to(DEV)
    cam.view_world_transform = None
    m = raw_model(30_000, 44)
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        got = renderer.render(cam, m, renderer.PipelineParams, bg)
    want = reference_shaped_render(cam, m, bg)
    for k in RENDER_KEYS:
        assert torch.equal(got[k], want[k]), k


def test_parameters_changed_every_frame_are_followed():
    """No memo, nothing cached: in-place edits between frames (training, dynamic scenes) -- also ones that bypass the
    autograd version counter -- show up in the next frame, and the result equals the reference-shaped render of the
    edited model."""
    cam = orbit_cameras(12, 240, 136)[1].to(DEV)
    m = raw_model(25_000, 55, nasty=False)
    bg = torch.tensor([0.1, 0.1, 0.1], device=DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    with torch.no_grad():
        prev = renderer.render(cam, m, renderer.PipelineParams, bg)["render"].clone()
        for step in range(3):
            m._xyz.data.add_(0.02 * torch.randn(m._xyz.shape, device=DEV, generator=g))      # version counter bypassed
            m._scaling.add_(0.05)
            m._opacity.mul_(0.9)
            m._rotation.add_(0.05 * torch.randn(m._rotation.shape, device=DEV, generator=g))
            m._features_dc.mul_(1.05)
            got = renderer.render(cam, m, renderer.PipelineParams, bg)
            want = reference_shaped_render(cam, m, bg)
            assert not torch.equal(got["render"], prev)
            for k in RENDER_KEYS:
                assert torch.equal(got[k], want[k]), (step, k)
            prev = got["render"].clone()


def test_render_begin_takes_the_raw_path():
    cam = orbit_cameras(12, 200, 120)[3].to(DEV)
    m = raw_model(15_000, 66)
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        pending = renderer.render_begin(cam, m, renderer.PipelineParams, bg)
        got = pending.finish()
        want = renderer.render(cam, m, renderer.PipelineParams, bg)
    for k in RENDER_KEYS:
        assert torch.equal(got[k], want[k]), k


def test_models_that_must_not_take_the_raw_path():
    """A swapped activation, an opt-out flag or autograd keep render() on the getters: still correct, only slower."""
    cam = orbit_cameras(12, 160, 96)[8].to(DEV)
    m = raw_model(5_000, 77, nasty=False)
    assert renderer.raw_parameters(m) is not None
    m.scaling_activation = lambda x: torch.exp(x) * 2.0
    assert renderer.raw_parameters(m) is None
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        doubled = renderer.render(cam, m, renderer.PipelineParams, bg)
    m.setup_functions()
    m.gsr_raw_parameters = False
    assert renderer.raw_parameters(m) is None
    with torch.no_grad():
        plain = renderer.render(cam, m, renderer.PipelineParams, bg)
    assert not torch.equal(doubled["radii"], plain["radii"])
    del m.gsr_raw_parameters
    with torch.no_grad():
        raw = renderer.render(cam, m, renderer.PipelineParams, bg)
    for k in RENDER_KEYS:
        assert torch.equal(raw[k], plain[k]), k


def test_memoising_model_and_raw_path_agree():
    """The builder's own GaussianModel (rounds 2-3: memoised getters + gsr_view_normals) and the raw path give the same
    frame: the stand-alone normal kernel was brought to the same PyTorch-exact arithmetic."""
    cam = orbit_cameras(12, 320, 180)[4].to(DEV)
    r = raw_model(30_000, 88)
    ours = gm.GaussianModel(3)
    ours._xyz, ours._scaling, ours._rotation, ours._opacity = r._xyz, r._scaling, r._rotation, r._opacity
    ours._features_dc, ours._features_rest, ours.active_sh_degree = r._features_dc, r._features_rest, 3
    bg = torch.tensor([0.5, 0.5, 0.5], device=DEV)
    saved = renderer.RAW_PARAMETERS
    try:
        with torch.no_grad():
            renderer.RAW_PARAMETERS = False
            memo = renderer.render(cam, ours, renderer.PipelineParams, bg)
            renderer.RAW_PARAMETERS = True
            raw = renderer.render(cam, ours, renderer.PipelineParams, bg)
    finally:
        renderer.RAW_PARAMETERS = saved
    for k in RENDER_KEYS:
        assert torch.equal(memo[k], raw[k]), k


@pytest.mark.parametrize("frame", [0, 400])
def test_c3_full_size_raw_render_is_bit_identical(frame):
    """BASELINE configs[2] at full size (3 M Gaussians, 1920x1080): the raw fused render against the reference-shaped one."""
    cloud = scenes.config_c3()
    cam = orbit_cameras(800, 1920, 1080)[frame].to(DEV)
    m = raw_model(0, 2, cloud=cloud, nasty=False)
    del cloud
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        got = renderer.render(cam, m, renderer.PipelineParams, bg)
    want = reference_shaped_render(cam, m, bg)
    for k in RENDER_KEYS:
        assert torch.equal(got[k], want[k]), k
    assert int(got["visibility_filter"].sum()) > 1_500_000 and math.isfinite(float(got["render"].sum()))
