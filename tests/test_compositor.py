"""Compositor: the numpy oracle against the reference's own blend_frames (run on synthetic layers with its
file loaders monkey-patched), then the HIP kernel against the oracle."""
import importlib
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import compositor_oracle as co

BLENDER = "/root/reference/blender"
needs_reference = pytest.mark.skipif(not os.path.isdir(BLENDER), reason="reference tree not mounted")


def synthetic_layers(H, W, seed, with_3dgs=False, with_smoke=False, with_fire=False):
    g = np.random.default_rng(seed)
    rgba = lambda a_sparsity: np.concatenate(
        (g.integers(0, 256, (H, W, 3)), (g.integers(0, 256, (H, W, 1)) * (g.random((H, W, 1)) > a_sparsity))), -1).astype(np.uint8)
    depth = lambda lo, hi: g.uniform(lo, hi, (H, W)).astype(np.float32)
    L = {"bg_c": rgba(0.0), "o_c": rgba(0.6), "o_d": depth(1, 6), "s_c": rgba(0.0), "s_d": depth(2, 5)}
    L["bg_c"][..., 3] = 255
    # object+shadow pass: the catcher darkened here and there
    o_s = L["s_c"].copy()
    dark = g.random((H, W)) > 0.7
    o_s[dark, :3] = (o_s[dark, :3] * g.uniform(0.2, 0.9, (int(dark.sum()), 1))).astype(np.uint8)
    o_s[..., 3] = (g.integers(0, 256, (H, W)) * (g.random((H, W)) > 0.3)).astype(np.uint8)
    L["o_s_c"] = o_s
    if with_3dgs:
        L["o_gs_c"], L["o_gs_d"] = rgba(0.5), depth(1, 6)
    if with_smoke:
        L["s_f_c"], L["s_f_d"] = rgba(0.5), depth(1, 6)
        if with_fire:
            L["s_f_c_pre"] = rgba(0.5)
    return L


def run_reference_blend_frames(layers, tmp_path):
    """Drive /root/reference/blender/blend_all.py::blend_frames with in-memory layers."""
    for missing in ("cv2", "imageio", "imageio.v2", "skimage", "skimage.transform"):
        sys.modules.setdefault(missing, types.ModuleType(missing))
    sys.modules["imageio"].v2 = sys.modules["imageio.v2"]
    if BLENDER not in sys.path:
        sys.path.insert(0, BLENDER)
    ba = importlib.import_module("blend_all")
    cache = tmp_path / "cache" / "out"
    (cache / "rgb_all").mkdir(parents=True)
    (cache / "rgb_all" / "001.png").write_bytes(b"")      # only its existence is counted (:127-128)
    results = tmp_path / "scene" / "custom_camera_path" / "traj" / "exp"
    results.mkdir(parents=True)
    cfg = tmp_path / "cfg.json"
    cfg.write_text('{"blender_cache_dir": "%s", "output_dir_name": "out"}' % str(tmp_path / "cache"))
    key_of = {"rgb_obj": "o_c", "rgb_shadow": "s_c", "rgb_all": "o_s_c", "rgb_obj_3dgs": "o_gs_c",
              "rgb_smoke_fire": "s_f_c", "rgb_smoke_fire_pre": "s_f_c_pre", "depth_obj": "o_d", "depth_shadow": "s_d",
              "depth_all": "s_d", "depth_obj_3dgs": "o_gs_d", "depth_smoke_fire": "s_f_d", "depth_smoke_fire_pre": "s_f_d"}

    def pick(path):
        if path == "BG":
            return layers["bg_c"]
        for part in path.split(os.sep):
            if part in key_of:
                v = layers.get(key_of[part])
                return None if v is None else v.copy()
        raise AssertionError(path)

    ba.load_rgb = lambda p: pick(p)
    ba.load_depth = lambda p: np.zeros(layers["bg_c"].shape[:2], np.float32)
    ba.load_depth_exr = lambda p: pick(p)
    ba.generate_video_from_frames = lambda *a, **k: None
    real_glob = ba.glob.glob
    ba.glob.glob = lambda pat: (["BG"] if pat.endswith(os.path.join("images", "*.png")) else
                                ["BGD"] if pat.endswith(os.path.join("depth", "*.npy")) else real_glob(pat))
    try:
        ba.blend_frames(str(results), str(cfg))
    finally:
        ba.glob.glob = real_glob
    from PIL import Image
    return np.array(Image.open(results / "frames" / "0000.png"))


@needs_reference
@pytest.mark.parametrize("variant", ["plain", "3dgs", "smoke", "fire", "all"])
def test_oracle_matches_reference_blend_frames(tmp_path, variant):
    L = synthetic_layers(37, 53, seed=hash(variant) % 1000, with_3dgs=variant in ("3dgs", "all"),
                         with_smoke=variant in ("smoke", "fire", "all"), with_fire=variant in ("fire", "all"))
    want = run_reference_blend_frames(L, tmp_path)
    args = dict(L)
    if "s_f_c" in args:
        args["s_f_d"], _ = co.smoke_depth_fill(args["s_f_c"], args["s_f_d"], None)
    got = co.composite_frame(**args)
    assert got.dtype == np.uint8 and got.shape == want.shape
    np.testing.assert_array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "3dgs", "smoke", "fire", "all"])
@pytest.mark.parametrize("hw", [(540, 960), (37, 53)])
def test_hip_compositor_matches_oracle(variant, hw):
    from autovfx_amd import compositor
    L = synthetic_layers(hw[0], hw[1], seed=7 + len(variant), with_3dgs=variant in ("3dgs", "all"),
                         with_smoke=variant in ("smoke", "fire", "all"), with_fire=variant in ("fire", "all"))
    args = dict(L)
    if "s_f_c" in args:
        args["s_f_d"], _ = co.smoke_depth_fill(args["s_f_c"], args["s_f_d"], None)
    want = co.composite_frame(**args)
    dev = "cuda:0"
    t = {k: torch.from_numpy(v).to(dev) for k, v in args.items()}
    got = compositor.composite_frame(**t)
    torch.cuda.synchronize()
    assert got.dtype == torch.uint8 and tuple(got.shape) == want.shape
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_percentile_rule_matches_numpy():
    from autovfx_amd.compositor import _percentile_linear
    g = np.random.default_rng(2)
    for n in (1, 2, 7, 1000, 50_001):
        v = g.uniform(0.5, 9.0, n).astype(np.float32)
        for q in (0.001, 0.1, 37.5, 99.999):
            assert float(_percentile_linear(torch.from_numpy(v), q)) == float(np.percentile(v, q)), (n, q)


@pytest.mark.gpu
def test_smoke_depth_fill_matches_oracle():
    from autovfx_amd import compositor
    L = synthetic_layers(120, 90, seed=4, with_smoke=True)
    want, _ = co.smoke_depth_fill(L["s_f_c"], L["s_f_d"], None)
    got = compositor.smoke_depth_fill(torch.from_numpy(L["s_f_c"]).cuda(), torch.from_numpy(L["s_f_d"]).cuda())
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_compositor_throughput_report():
    """Not a pass/fail bar: records the kernel's achieved bandwidth next to its algorithmic bytes."""
    from autovfx_amd import compositor
    from test_parity_gpu import report
    H, W = 1080, 1920
    L = synthetic_layers(H, W, seed=1, with_3dgs=True, with_smoke=True, with_fire=True)
    t = {k: torch.from_numpy(v).cuda() for k, v in L.items()}
    out = torch.empty(H, W, 4, dtype=torch.uint8, device="cuda")
    for _ in range(5):
        compositor.composite_frame(out=out, **t)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        compositor.composite_frame(out=out, **t)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    alg = H * W * (7 * 4 + 4 * 4 + 4)     # 7 RGBA8 layers + 4 depth maps in, RGBA8 out
    report("compositor_1080p_all_layers", ms_per_frame=ms, alg_bytes=alg, alg_GBps=alg / (ms * 1e-3) / 1e9)
    assert ms < 5.0
