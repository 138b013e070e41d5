"""Poisoned-scratch suite: results must not depend on what freshly allocated memory contains.

Every tensor the binding hands to the library -- the three scratch arenas, the output images, radii, the gradient
tensors and the backward's accumulation scratch -- is ``torch.empty``.  A cold process gets zeroed pages from the
driver, a warm one gets whatever the caching allocator recycled; the library's contract is that it writes everything it
later reads (one zero-filled block per call, ``gsr_api.hip``; every other array is fully defined by the kernel that
owns it).  Here each entry point runs with the memory pre-filled with zeros, with 0xFF bytes (every float a NaN, every
index 4 G) and with random bytes (``_C.set_alloc_poison``), and must give the zero-prefilled run's results bit for bit:
forward (full call with every scratch sub-array decoded; inference call with the default policy and forced into
slabs), ``gsr_blend`` over cached geometry, ``gsr_forward_extra``, and the backward (integers exact, gradients -- sums of
float atomics whose order varies from run to run -- to 1e-5 of their scale, and finite).
"""
import numpy as np
import pytest
import torch

from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras

from helpers import hip_forward_inference, hip_forward_raw, settings_for

pytestmark = pytest.mark.gpu
PATTERNS = (0xFF, "random", 0x7F)


def _scene(name):
    if name == "c1":
        return scenes.config_c1(), scenes.c1_camera()
    if name == "ragged":
        return scenes.config_c1(P=777, seed=21), scenes.c1_camera(250, 130)
    if name == "heavy15k":
        return scenes.config_heavy(P=15_000), orbit_cameras(200, 960, 540)[3]
    if name == "c2":
        return scenes.config_c2(), orbit_cameras(200, 960, 540)[100]
    raise KeyError(name)


@pytest.fixture(autouse=True)
def _reset_poison():
    from diff_gaussian_rasterization import _C
    yield
    _C.set_alloc_poison(None)


def _with_poison(pattern, fn):
    from diff_gaussian_rasterization import _C
    _C.set_alloc_poison(pattern)
    try:
        out = fn()
        torch.cuda.synchronize()
        return out
    finally:
        _C.set_alloc_poison(None)


INT_KEYS = ("radii", "depth_order", "point_offsets", "tiles_touched", "point_list", "tile_keys", "ranges", "n_contrib",
            "live_mask", "tight_rect")
FLOAT_KEYS = ("color", "depth", "alpha")


@pytest.mark.parametrize("scene", ["c1", "ragged", "heavy15k", "c2"])
def test_full_call_every_array(scene):
    cloud, cam = _scene(scene)
    base = _with_poison(0, lambda: hip_forward_raw(cloud, cam, cull=True, bg=(0.1, 0.2, 0.3)))
    vis = base["radii"] > 0
    for pat in PATTERNS:
        got = _with_poison(pat, lambda: hip_forward_raw(cloud, cam, cull=True, bg=(0.1, 0.2, 0.3)))
        assert got["num_rendered"] == base["num_rendered"] and got["live_pairs"] == base["live_pairs"]
        for k in INT_KEYS:   # (the splat records are defined for every Gaussian: zeros when nothing is emitted)
            np.testing.assert_array_equal(got[k], base[k], err_msg=f"{scene} poison {pat}: {k}")
        for k in FLOAT_KEYS:
            np.testing.assert_array_equal(got[k].view(np.uint32), base[k].view(np.uint32), err_msg=f"{scene} poison {pat}: {k}")
        for k in ("depths", "means2D", "conic_opacity"):   # per-Gaussian records exist where the Gaussian is visible
            np.testing.assert_array_equal(got[k][vis].view(np.uint32), base[k][vis].view(np.uint32), err_msg=f"{scene} poison {pat}: {k}")
        listed = np.unique(base["point_list"])
        np.testing.assert_array_equal(got["rgb"][listed].view(np.uint32), base["rgb"][listed].view(np.uint32),
                                      err_msg=f"{scene} poison {pat}: rgb of listed Gaussians")


@pytest.mark.parametrize("scene", ["c1", "ragged", "heavy15k", "c2"])
@pytest.mark.parametrize("mode", ["policy", "two_slabs", "many_slabs", "one_slab"])
def test_inference_call(scene, mode):
    cloud, cam = _scene(scene)
    opts = {"policy": dict(slab_min_rest=3_000_000), "two_slabs": dict(slabs=2, slab_first=12), "many_slabs": dict(slabs=0, slab_first=6),
            "one_slab": dict(slabs=1)}[mode]
    run = lambda: hip_forward_inference(cloud, cam, bg=(0.3, 0.1, 0.2), **opts)
    base = _with_poison(0, run)
    for pat in PATTERNS:
        got = _with_poison(pat, run)
        assert got["num_rendered"] == base["num_rendered"] and got["slab_pairs"] == base["slab_pairs"], (scene, mode, pat)
        np.testing.assert_array_equal(got["radii"], base["radii"])
        for k in FLOAT_KEYS:
            np.testing.assert_array_equal(got[k].view(np.uint32), base[k].view(np.uint32), err_msg=f"{scene} {mode} poison {pat}: {k}")


@pytest.mark.parametrize("scene", ["c1", "ragged", "heavy15k", "c2"])
@pytest.mark.parametrize("inference", [False, True])
def test_second_blend_and_fused_second_feature(scene, inference):
    """gsr_blend over the scratch of an earlier call (the geometry cache's second pass) and gsr_forward_extra."""
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import _C
    dev = "cuda:0"
    cloud, cam = _scene(scene)
    c = cloud.to(dev)
    st = settings_for(cam, dev, (0.2, 0.4, 0.1), 1.0, cloud.sh_degree)
    e = torch.Tensor([])
    extra = torch.rand((cloud.P, 3), generator=torch.Generator(device=dev).manual_seed(3), device=dev)

    args = lambda colors, sh: (st.bg, c.means3D, colors, c.opacities, c.scales, c.rotations, 1.0, e, st.viewmatrix, st.projmatrix,
                               st.tanfovx, st.tanfovy, st.image_height, st.image_width, sh, st.sh_degree, st.campos, False, False)

    def run():
        _lib.set_option(_lib.OPT_SLABS, 0); _lib.set_option(_lib.OPT_SLAB_FIRST, 12); _lib.set_option(_lib.OPT_SLAB_MIN_REST, 0)
        try:
            with torch.no_grad():
                _C.set_geometry_cache(True)   # second call: same geometry tensors, new colours -> gsr_blend over the first's lists
                h0 = _C.cache_stats["hits"]
                a = _C.rasterize_gaussians(*args(e, c.shs), inference=inference)
                b = _C.rasterize_gaussians(*args(extra, e), inference=inference)
                assert _C.cache_stats["hits"] == h0 + 1
                outs = [a[i].clone() for i in (1, 2, 3, 4)] + [b[i].clone() for i in (1, 2, 3, 4)]
                _C.set_geometry_cache(False)
                fused = _C.rasterize_gaussians_extra(*args(e, c.shs), extra, inference=inference)
                outs += [fused[i].clone() for i in (1, 2, 3, 4, 8)]
            torch.cuda.synchronize()
            return outs
        finally:
            _C.set_geometry_cache(None)
            _lib.set_option(_lib.OPT_SLABS, 2); _lib.set_option(_lib.OPT_SLAB_FIRST, 400); _lib.set_option(_lib.OPT_SLAB_MIN_REST, 3_000_000)

    base = _with_poison(0, run)
    for pat in PATTERNS:
        got = _with_poison(pat, run)
        for i, (x, y) in enumerate(zip(got, base)):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32)), (scene, inference, pat, i)


@pytest.mark.parametrize("scene", ["c1", "ragged", "heavy15k", "c2"])
def test_backward(scene):
    from test_backward_gpu import KEYS_SH, hip_backward
    from test_oracle_backward import pixel_grads
    cloud, cam = _scene(scene)
    pg = pixel_grads(cam, 5)
    base = _with_poison(0, lambda: hip_backward(cloud, cam, pg, bg=(0.1, 0.2, 0.3)))
    again = _with_poison(0, lambda: hip_backward(cloud, cam, pg, bg=(0.1, 0.2, 0.3)))   # run-to-run wobble of the float atomics
    for pat in PATTERNS:
        got = _with_poison(pat, lambda: hip_backward(cloud, cam, pg, bg=(0.1, 0.2, 0.3)))
        np.testing.assert_array_equal(got["radii"], base["radii"])
        np.testing.assert_array_equal(got["color"].view(np.uint32), base["color"].view(np.uint32))
        for k in KEYS_SH:
            a, b = got[k].astype(np.float64), base[k].astype(np.float64)
            assert np.isfinite(a).all(), f"{scene} poison {pat}: {k} has non-finite values"
            scale = float(np.abs(b).max())
            wobble = float(np.abs(again[k].astype(np.float64) - b).max())
            assert float(np.abs(a - b).max()) <= 1e-5 * scale + 4 * wobble + 1e-12, f"{scene} poison {pat}: {k}"
