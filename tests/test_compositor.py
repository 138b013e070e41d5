"""Compositor: the numpy oracle against the reference's own blend_frames (run on synthetic layers with its
file loaders monkey-patched), then the HIP kernel against the oracle."""
import importlib
import json
import os
import struct
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


# ---- the input side: Blender layers at render resolution, brought to the frame's size as blend_all.py:217-234 does ------------

def pil_downsample(image, new_size):
    """blend_all.py:21-28, verbatim semantics (PIL is the oracle here)."""
    from PIL import Image
    img = Image.fromarray(image)
    return np.array(img.resize(new_size, resample=Image.BILINEAR) if image.ndim == 3 else img.resize(new_size, Image.NEAREST))


def oracle_from_blender_layers(L, hw):
    """The reference's order of operations on layers that arrive at Blender's resolution: smoke depth fill on the full-size
    layers (:207-215), then every layer through downsample_image (:217-234), then the per-pixel composite (:236-343)."""
    args = dict(L)
    if "s_f_c" in args:
        args["s_f_d"], _ = co.smoke_depth_fill(args["s_f_c"], args["s_f_d"], None)
    new_size = (hw[1], hw[0])
    for k in args:
        if k != "bg_c":
            args[k] = pil_downsample(args[k], new_size)
    return co.composite_frame(**args)


def blender_layers(hw, scale_hw, seed, **kw):
    """Background at the frame's size, every Blender layer at ``scale_hw`` times it."""
    big = synthetic_layers(int(hw[0] * scale_hw[0]), int(hw[1] * scale_hw[1]), seed, **kw)
    big["bg_c"] = synthetic_layers(hw[0], hw[1], seed + 1)["bg_c"]
    return big


@needs_reference
@pytest.mark.parametrize("variant,scale", [("plain", (2, 2)), ("all", (2, 2)), ("fire", (1.5, 1.25))])
def test_oracle_with_pil_resizes_matches_reference_blend_frames(tmp_path, variant, scale):
    """The reference's own blend_frames on Blender layers at 2x (its anti-aliasing setup) and at an odd ratio: PIL resizes
    inside it; the oracle path above must give the same frame."""
    hw = (36, 52)
    L = blender_layers(hw, scale, seed=11, with_3dgs=variant == "all", with_smoke=variant in ("fire", "all"), with_fire=variant in ("fire", "all"))
    want = run_reference_blend_frames(L, tmp_path)
    np.testing.assert_array_equal(oracle_from_blender_layers(L, hw), want)


@pytest.mark.gpu
@pytest.mark.parametrize("src_hw,dst_hw", [((1080, 1920), (540, 960)), ((216, 384), (108, 192)), ((100, 150), (33, 77)), ((37, 53), (37, 20)),
                                           ((50, 60), (57, 60)), ((64, 64), (128, 100)), ((31, 17), (31, 17)), ((5, 7), (1, 1)), ((1, 1), (4, 6)),
                                           ((270, 481), (135, 240))])
def test_gpu_resizes_equal_pillow_bit_for_bit(src_hw, dst_hw):
    """gsr_resize_rgba8_bilinear / gsr_resize_f32_nearest against Pillow itself: 2x down (the reference's case), odd ratios,
    one dimension only (the other pass is skipped but the premultiply round trip stays), up-scaling, the identity (a copy), to
    and from one pixel.  Alpha is 0 / 255 on a third of the pixels each (the un-premultiply's special cases)."""
    from autovfx_amd import compositor
    g = np.random.default_rng(src_hw[0] * 7 + dst_hw[1])
    img = g.integers(0, 256, src_hw + (4,), dtype=np.uint8)
    img[g.random(src_hw) < 0.33, 3] = 0
    img[g.random(src_hw) < 0.33, 3] = 255
    depth = g.uniform(0.5, 30.0, src_hw).astype(np.float32)
    new_size = (dst_hw[1], dst_hw[0])
    got_c = compositor.resize_rgba8(torch.from_numpy(img).cuda(), new_size)
    got_d = compositor.resize_depth(torch.from_numpy(depth).cuda(), new_size)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got_c.cpu().numpy(), pil_downsample(img, new_size))
    np.testing.assert_array_equal(got_d.cpu().numpy(), pil_downsample(depth, new_size))


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "3dgs", "all"])
@pytest.mark.parametrize("hw,scale", [((540, 960), (2, 2)), ((90, 160), (1.5, 1.25))])
def test_hip_compositor_takes_blender_resolution_layers(variant, hw, scale):
    """composite_frame fed the Blender layers as Blender renders them (2x the frame, or any other size): resized on the GPU
    like blend_all.py:217-234, composited, and equal to the oracle (PIL resizes + numpy composite) byte for byte."""
    from autovfx_amd import compositor
    L = blender_layers(hw, scale, seed=21 + len(variant), with_3dgs=variant in ("3dgs", "all"), with_smoke=variant == "all", with_fire=variant == "all")
    want = oracle_from_blender_layers(L, hw)
    t = {k: torch.from_numpy(v).cuda() for k, v in L.items()}
    if "s_f_c" in t:
        t["s_f_d"] = compositor.smoke_depth_fill(t["s_f_c"], t["s_f_d"])     # on the full-size layers, before the resize (:207-215)
    got = compositor.composite_frame(**t)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def write_blender_tree(root, L, hw, frames=2, half=False, compression="ZIP"):
    """The directory layout blend_frames expects (blend_all.py:118-182), filled with synthetic layers: the 3DGS frames under
    <scene>/custom_camera_path/images, Blender's passes under <cache>/out/{rgb,depth}_*; the depth passes are real OpenEXR files as
    Blender's File Output node writes them -- R, G, B, A channels all holding the Z pass, ZIP compression (autovfx_amd.exr.write_exr);
    ``half``: 16-bit floats, Blender's default colour depth (the layers handed back are then the values such a file holds)."""
    from PIL import Image
    from autovfx_amd import exr
    results = root / "scene" / "custom_camera_path" / "traj" / "exp"
    results.mkdir(parents=True)
    images = root / "scene" / "custom_camera_path" / "images"
    images.mkdir(parents=True)
    cache = root / "cache" / "out"
    cfg = root / "cfg.json"
    cfg.write_text('{"blender_cache_dir": "%s", "output_dir_name": "out"}' % str(root / "cache"))
    kinds = {"rgb_obj": "o_c", "rgb_shadow": "s_c", "rgb_all": "o_s_c", "rgb_obj_3dgs": "o_gs_c", "rgb_smoke_fire": "s_f_c", "rgb_smoke_fire_pre": "s_f_c_pre"}
    depths = {"depth_obj": "o_d", "depth_shadow": "s_d", "depth_all": "s_d", "depth_obj_3dgs": "o_gs_d", "depth_smoke_fire": "s_f_d"}   # (depth_all is loaded and resized by the reference, then never used)
    per_frame = []
    for i in range(frames):
        Li = {k: (np.roll(v, 3 * i, axis=1) if isinstance(v, np.ndarray) else v) for k, v in L.items()}
        if half:
            Li = {k: (v.astype(np.float16).astype(np.float32) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in Li.items()}
        Image.fromarray(Li["bg_c"]).save(images / f"{i:05d}.png")
        for kind, key in kinds.items():
            if key in Li:
                (cache / kind).mkdir(parents=True, exist_ok=True)
                Image.fromarray(Li[key]).save(cache / kind / f"{i + 1:03d}.png")
        for kind, key in list(depths.items()) + ([("depth_smoke_fire_pre", "s_f_d")] if "s_f_c_pre" in Li else []):   # (resized by the reference with the fire layer, never used)
            if key in Li:
                d = cache / kind / f"{i + 1:03d}"
                d.mkdir(parents=True, exist_ok=True)
                z = Li[key]
                exr.write_exr(str(d / f"Image{i + 1:04d}.exr"), {"R": z, "G": z, "B": z, "A": np.ones_like(z)}, compression=compression, half=half)
        per_frame.append(Li)
    return results, cfg, per_frame


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plain", "all", "all-half-zips"])
def test_blend_frames_drop_in_on_a_directory_tree(tmp_path, variant):
    """autovfx_amd.compositor.blend_frames -- what ``blend_all.blend_frames`` becomes after ``autovfx_amd.install()`` -- on a
    directory tree shaped like the reference's (blend_all.py:118-182), Blender layers at 2x the frame: the frames it writes decode
    (PIL) to the oracle's composite of the same layers, byte for byte; the frame count follows rgb_all/*.png."""
    from PIL import Image
    from autovfx_amd import compositor
    hw = (54, 96)
    full = variant != "plain"
    L = blender_layers(hw, (2, 2), seed=5, with_3dgs=full, with_smoke=full, with_fire=full)
    results, cfg, per_frame = write_blender_tree(tmp_path, L, hw, frames=5, half=variant.endswith("zips"),
                                                 compression="ZIPS" if variant.endswith("zips") else "ZIP")
    paths = compositor.blend_frames(str(results), str(cfg), write_video=False)     # (EXR depth passes read by autovfx_amd.exr: no OpenCV here)
    assert [os.path.basename(p) for p in paths] == [f"{i:04d}.png" for i in range(5)]
    for path, Li in zip(paths, per_frame):
        got = np.array(Image.open(path))
        np.testing.assert_array_equal(got, oracle_from_blender_layers(Li, hw), err_msg=path)


@pytest.mark.gpu
def test_blend_frames_hands_the_frames_to_the_video_writer_and_survives_its_fallbacks(tmp_path, monkeypatch):
    """The branches a plain run does not take: with ``imageio`` / ``skimage`` importable (doubles here) every frame also comes back to
    the host, in order, for ``blended.mp4`` (blend_all.py:31-54); more pool threads than frames; the EXR streams inflated by zlib on
    the host (``AUTOVFX_AMD_EXR_INFLATE=host``) and a depth pass the GPU's decoder refuses (the frame is read again on the host) give
    the same frames."""
    from PIL import Image
    from autovfx_amd import compositor
    hw = (54, 96)
    L = blender_layers(hw, (2, 2), seed=7, with_3dgs=True, with_smoke=True, with_fire=True)
    results, cfg, per_frame = write_blender_tree(tmp_path, L, hw, frames=7, half=True, compression="ZIP")
    want = [oracle_from_blender_layers(Li, hw) for Li in per_frame]
    seen = {}
    imageio, v2, skimage, transform = (types.ModuleType(n) for n in ("imageio", "imageio.v2", "skimage", "skimage.transform"))
    v2.mimsave = lambda path, series, fps, macro_block_size: seen.update(path=path, series=[np.array(f) for f in series], fps=fps)
    transform.resize = lambda frame, shape: frame.astype(np.float64) / 255.0 if tuple(frame.shape[:2]) == tuple(shape) else None
    imageio.v2, skimage.transform = v2, transform
    for name, mod in (("imageio", imageio), ("imageio.v2", v2), ("skimage", skimage), ("skimage.transform", transform)):
        monkeypatch.setitem(sys.modules, name, mod)
    monkeypatch.setenv("AUTOVFX_AMD_BLEND_DECODERS", "12")                      # more threads than frames
    paths = compositor.blend_frames(str(results), str(cfg))
    assert seen["path"].endswith("blended.mp4") and seen["fps"] == 15 and len(seen["series"]) == 7
    for k, (path, frame) in enumerate(zip(paths, seen["series"])):
        np.testing.assert_array_equal(np.array(Image.open(path)), want[k], err_msg=path)
        np.testing.assert_array_equal(frame, want[k], err_msg=f"video frame {k}")
    monkeypatch.setenv("AUTOVFX_AMD_BLEND_DECODERS", "2")
    monkeypatch.setenv("AUTOVFX_AMD_EXR_INFLATE", "host")
    for path, w in zip(compositor.blend_frames(str(results), str(cfg), write_video=False), want):
        np.testing.assert_array_equal(np.array(Image.open(path)), w, err_msg=path + " (host inflate)")
    # a flipped bit inside the first zlib stream of frame 3's object depth pass: the GPU's decoder refuses it; zlib on the host does too,
    # and the file then goes the way of every file the kernels do not cover -- the host reader, which raises
    monkeypatch.setenv("AUTOVFX_AMD_EXR_INFLATE", "gpu")
    cache = os.path.join(json.load(open(cfg))["blender_cache_dir"], json.load(open(cfg))["output_dir_name"])
    victim = os.path.join(cache, "depth_obj", "004", "Image0004.exr")
    from autovfx_amd import exr
    buf = bytearray(open(victim, "rb").read())
    first = struct.unpack_from("<Q", buf, exr.read_header(bytes(buf))["offsets_at"])[0]
    buf[first + 8 + 12] ^= 0x20
    open(victim, "wb").write(bytes(buf))
    with pytest.raises(Exception):
        compositor.blend_frames(str(results), str(cfg), write_video=False)


@needs_reference
def test_reference_blend_frames_on_the_same_directory_tree(tmp_path):
    """The reference's own blend_frames (its file discovery, PIL loaders and resizes; only the EXR loader and the video writer
    patched) on the tree ``write_blender_tree`` writes: its frames equal the oracle path frame by frame -- which is what the GPU
    drop-in is compared with on the GPU box, where the reference tree does not exist."""
    from PIL import Image
    for missing in ("cv2", "imageio", "imageio.v2", "skimage", "skimage.transform"):
        sys.modules.setdefault(missing, types.ModuleType(missing))
    sys.modules["imageio"].v2 = sys.modules["imageio.v2"]
    if BLENDER not in sys.path:
        sys.path.insert(0, BLENDER)
    ba = importlib.import_module("blend_all")
    ba = importlib.reload(ba)     # (another test may have patched its loaders)
    hw = (54, 96)
    L = blender_layers(hw, (2, 2), seed=5, with_3dgs=True, with_smoke=True, with_fire=True)
    results, cfg, per_frame = write_blender_tree(tmp_path, L, hw)
    (tmp_path / "scene" / "custom_camera_path" / "depth").mkdir()
    for i in range(len(per_frame)):
        np.save(tmp_path / "scene" / "custom_camera_path" / "depth" / f"{i:05d}.npy", np.zeros(hw, np.float32))
    from autovfx_amd import exr
    ba.load_depth_exr = lambda p: exr.load_depth_exr(p) if os.path.exists(p) else None    # (the reference's is two lines around cv2.imread, which this image lacks)
    ba.generate_video_from_frames = lambda *a, **k: None
    ba.blend_frames(str(results), str(cfg))
    for i, Li in enumerate(per_frame):
        got = np.array(Image.open(results / "frames" / f"{i:04d}.png"))
        np.testing.assert_array_equal(got, oracle_from_blender_layers(Li, hw), err_msg=f"frame {i}")
