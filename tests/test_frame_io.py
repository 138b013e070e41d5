"""Frame writers: dependency-free PNG against PIL's decoder, and the four per-frame files."""
import io

import numpy as np
import pytest
import torch

from autovfx_amd import frame_io


@pytest.mark.parametrize("shape", [(7, 5, 4), (64, 33, 3), (1, 1, 4)])
def test_png_roundtrip_and_pil_agrees(shape):
    img = np.random.default_rng(1).integers(0, 256, shape).astype(np.uint8)
    data = frame_io.encode_png(img)
    np.testing.assert_array_equal(frame_io.decode_png(data), img)
    PIL = pytest.importorskip("PIL.Image")
    np.testing.assert_array_equal(np.array(PIL.open(io.BytesIO(data))), img)


def test_write_frame_outputs(tmp_path):
    g = torch.Generator().manual_seed(0)
    H, W = 12, 20
    result = {"render": torch.rand(4, H, W, generator=g), "depth": torch.rand(H, W, generator=g) * 5,
              "normal": torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)}
    paths = frame_io.write_frame_outputs(str(tmp_path), "00007", result)
    rgba = frame_io.decode_png(open(paths["images"], "rb").read())
    want = (result["render"] * 255 + 0.5).clamp(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    np.testing.assert_array_equal(rgba, want)
    np.testing.assert_array_equal(np.load(paths["depth"]), result["depth"].numpy())
    n = frame_io.decode_png(open(paths["normal"], "rb").read())
    np.testing.assert_array_equal(n, ((result["normal"] + 1) / 2 * 255).to(torch.uint8).numpy())
    assert paths["images"].endswith("images/00007.png") and paths["depth"].endswith("depth/00007.npy")
    # the depth preview: depth2img(depth, scale=3.0) through the turbo table (scene_representation.py:432-433)
    prev = frame_io.decode_png(open(paths["depth_preview"], "rb").read())
    idx = (np.clip(result["depth"].numpy() / 3.0, 0.0, 1.0) * 255).astype(np.uint8)
    np.testing.assert_array_equal(prev, frame_io.TURBO_LUT[idx])
    assert paths["depth_preview"].endswith("depth/00007.png") and prev.shape == (H, W, 3)


def test_turbo_table_is_the_published_one():
    """TURBO_LUT against matplotlib's copy of the turbo float table (the numbers OpenCV's colormap.cpp carries), x 255,
    rounded to nearest; end points and a mid entry typed in from cv2.applyColorMap's documented output (BGR reversed)."""
    lut = frame_io.TURBO_LUT
    assert lut.shape == (256, 3) and lut.dtype == np.uint8
    np.testing.assert_array_equal(lut[0], (48, 18, 59))      # dark blue
    np.testing.assert_array_equal(lut[255], (122, 4, 3))     # dark red
    assert lut[128, 1] > 240 and lut[128, 0] > 120 and lut[128, 2] < 100   # the green-yellow middle
    cm = pytest.importorskip("matplotlib._cm_listed")
    data = np.array(cm._turbo_data, dtype=np.float32)
    np.testing.assert_array_equal(lut, np.rint(data * np.float32(255.0)).astype(np.uint8))
    # depth2img: scale, clip, truncation to 8 bits
    d = np.array([-1.0, 0.0, 1.5, 2.999, 3.0, 40.0], np.float32)
    np.testing.assert_array_equal(frame_io.depth2img(d, 3.0), lut[[0, 0, 127, 254, 255, 255]])


def test_frame_writer_pool_writes_the_same_files(tmp_path):
    """FrameWriter (encoding on host threads behind the loop) leaves byte-identical files; errors surface in close()."""
    g = torch.Generator().manual_seed(1)
    H, W = 18, 31
    frames = [{"render": torch.rand(4, H, W, generator=g), "depth": torch.rand(H, W, generator=g) * 5,
               "normal": torch.nn.functional.normalize(torch.randn(H, W, 3, generator=g), dim=-1)} for _ in range(9)]
    a, b = tmp_path / "serial", tmp_path / "pool"
    for i, fr in enumerate(frames):
        frame_io.write_frame_outputs(str(a), f"{i:05d}", fr)
    with frame_io.FrameWriter(str(b), workers=3, max_pending=2) as w:
        for i, fr in enumerate(frames):
            w.submit(f"{i:05d}", fr)
            fr["render"].zero_()          # the frame was copied out: the caller may reuse its tensors at once
    for sub in ("images", "depth", "normal"):
        names = sorted(p.name for p in (a / sub).iterdir())
        assert names == sorted(p.name for p in (b / sub).iterdir()) and len(names) == (18 if sub == "depth" else 9)   # .npy + preview .png
        for n in names:
            assert (a / sub / n).read_bytes() == (b / sub / n).read_bytes(), (sub, n)
    w = frame_io.FrameWriter(str(tmp_path / "bad"), workers=1)
    w.submit("x", {"render": torch.rand(4, 4, 4), "depth": torch.rand(4, 4), "normal": torch.rand(4, 4, 2)})   # 2 channels: no PNG
    with pytest.raises(ValueError):
        w.close()


@pytest.mark.gpu
def test_render_trajectory_script_end_to_end(tmp_path):
    """PLY + trajectory JSON in, the reference's four per-frame files out (scripts/render_trajectory.py)."""
    import json
    import subprocess
    import sys
    from autovfx_amd import cameras, scenes
    from autovfx_amd.gaussian_model import GaussianModel
    c = scenes.config_c2(P=20_000, seed=3)
    ply = str(tmp_path / "point_cloud.ply")
    GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3).save_ply(ply)
    poses = cameras.orbit_c2w(4.0, 3)
    fx = cameras.fov2focal(np.deg2rad(60.0), 160)
    traj = str(tmp_path / "traj.json")
    with open(traj, "w") as f:
        json.dump(cameras.trajectory_dict("t", poses, fx, fx, 80, 45, 160, 90), f)
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, __import__("os").path.join(root, "scripts", "render_trajectory.py"), "--ply", ply,
                        "--trajectory", traj, "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    names = sorted(p.name for p in (out / "images").iterdir())
    assert names == ["00000.png", "00001.png", "00002.png"]
    rgba = frame_io.decode_png((out / "images" / "00001.png").read_bytes())
    depth = np.load(out / "depth" / "00001.npy")
    normal = frame_io.decode_png((out / "normal" / "00001.png").read_bytes())
    assert rgba.shape == (90, 160, 4) and depth.shape == (90, 160) and normal.shape == (90, 160, 3)
    assert rgba[..., 3].max() > 200 and depth.max() > 1.0          # something was rendered


@pytest.mark.gpu
def test_render_trajectory_script_with_moving_objects(tmp_path):
    """The same script with an inserted object that moves (--object, --rigid-body-json: the reference's rb_transform_info):
    a frame without an entry renders the base scene alone, a frame with one differs from it where the object is."""
    import json
    import subprocess
    import sys
    from autovfx_amd import cameras, scenes
    from autovfx_amd.gaussian_model import GaussianModel
    os_ = __import__("os")
    c = scenes.config_c2(P=20_000, seed=3)
    ply = str(tmp_path / "point_cloud.ply")
    GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3).save_ply(ply)
    o = scenes.config_c1(P=3000, seed=9)
    obj = str(tmp_path / "object_gaussians.ply")
    GaussianModel.from_activated(o.means3D * 0.3, o.opacities.clamp(min=0.6), o.scales * 0.6, o.rotations, o.shs, 3).save_ply(obj)
    poses = [cameras.orbit_c2w(4.0, 3)[0]] * 3      # one camera, three frames: only the object changes
    fx = cameras.fov2focal(np.deg2rad(60.0), 160)
    traj = str(tmp_path / "traj.json")
    with open(traj, "w") as f:
        json.dump(cameras.trajectory_dict("t", poses, fx, fx, 80, 45, 160, 90), f)
    rb = {"obj1": {"002": {"pos": [0.2, 0.1, 0.0], "rot": np.eye(3).tolist(), "scale": 1.0},
                   "003": {"pos": [-0.4, 0.3, 0.2], "rot": [[0, -1, 0], [1, 0, 0], [0, 0, 1]], "scale": 1.5}}}
    rbj = str(tmp_path / "rb.json")
    with open(rbj, "w") as f:
        json.dump(rb, f)
    root = os_.path.dirname(os_.path.dirname(os_.path.abspath(__file__)))
    outs = {}
    for name, extra in (("static", []), ("moving", ["--object", f"obj1={obj}@0,0,0", "--rigid-body-json", rbj])):
        out = tmp_path / name
        r = subprocess.run([sys.executable, os_.path.join(root, "scripts", "render_trajectory.py"), "--ply", ply, "--trajectory", traj,
                            "--out", str(out), *extra], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = [frame_io.decode_png((out / "images" / f"{i:05d}.png").read_bytes()) for i in range(3)]
    assert np.array_equal(outs["moving"][0], outs["static"][0]), "frame 001 has no object: the base scene alone"
    for i in (1, 2):
        assert int((outs["moving"][i].astype(int) - outs["static"][i].astype(int)).__abs__().max()) > 20, f"frame {i + 1}: the object is missing"
    assert not np.array_equal(outs["moving"][1], outs["moving"][2])


# ---- file images built on the GPU (gsr_png_encode, GpuFrameWriter) -----------------------------------------------------

def _paeth_stream(img: np.ndarray) -> np.ndarray:
    """The scanline stream of ``img`` with every row Paeth-filtered (PNG filter type 4), [H, 1 + W*C] uint8."""
    H, W, C = img.shape
    raw = img.reshape(H, W * C).astype(np.int16)
    left, up, ul = np.zeros_like(raw), np.zeros_like(raw), np.zeros_like(raw)
    left[:, C:] = raw[:, :-C]
    up[1:] = raw[:-1]
    ul[1:, C:] = raw[:-1, :-C]
    p = left + up - ul
    pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
    return np.concatenate((np.full((H, 1), 4, np.uint8), ((raw - pred) & 255).astype(np.uint8)), axis=1)


def _check_png_file(data: bytes, img: np.ndarray, paeth: bool = False):
    """A strict reader: every chunk's CRC against zlib's, the IDAT payload through zlib (which verifies the Adler-32), the
    scanlines against the image, and PIL's decoder on the whole file."""
    import struct
    import zlib
    h, w, c = img.shape
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, tags, idat = 8, [], b""
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + body) & 0xFFFFFFFF, f"CRC of chunk {tag!r}"
        tags.append(tag)
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    assert pos == len(data) and tags == [b"IHDR", b"IDAT", b"IEND"]
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    if paeth:
        np.testing.assert_array_equal(rows, _paeth_stream(img))
    else:
        assert not rows[:, 0].any()
        np.testing.assert_array_equal(rows[:, 1:].reshape(h, w, c), img)
    PIL = pytest.importorskip("PIL.Image")
    im = PIL.open(io.BytesIO(data))
    im.load()
    assert im.mode == ("RGBA" if c == 4 else "RGB")
    np.testing.assert_array_equal(np.asarray(im), img)
    if not paeth or img.size <= 100_000:      # (the package's own small reader un-filters Paeth rows in a Python loop)
        np.testing.assert_array_equal(frame_io.decode_png(data), img)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1, 4), (1, 1, 3), (7, 5, 4), (64, 33, 3), (3, 5461, 4), (3, 5462, 4), (300, 100, 4), (540, 960, 4),
                                   (540, 960, 3), (1080, 1920, 4), (2, 21845, 3), (17, 1285, 3)])
@pytest.mark.parametrize("planar", [False, True])
def test_gpu_png_files_are_valid_and_decode_to_the_image(shape, planar):
    """gsr_png_encode: the file bytes come off the GPU finished -- signature, IHDR, one IDAT of stored deflate blocks with its
    Adler-32, the chunk CRC, IEND.  Sizes around the 65535-byte block boundary (a scanline stream of exactly one block, one byte
    more, rows that straddle blocks), one pixel, the bench sizes; interleaved and planar sources."""
    h, w, c = shape
    img = np.random.default_rng(h * 131 + w + c).integers(0, 256, shape).astype(np.uint8)
    src = torch.from_numpy(img).cuda()
    if planar:
        src = src.permute(2, 0, 1).contiguous()
    out = frame_io.encode_png_gpu(src, planar=planar)
    torch.cuda.synchronize()
    assert out.numel() == frame_io.png_size(w, h, c)
    _check_png_file(out.cpu().numpy().tobytes(), img)


@pytest.mark.gpu
def test_gpu_png_of_constant_images_and_rejects_what_it_cannot_encode():
    """All-zero and all-255 images (Adler-32 sums at their extremes: s2 of a 1080p RGBA frame of 255s passes 1e15 before the
    modulus), and the argument errors."""
    for value in (0, 255):
        img = np.full((1080, 1920, 4), value, np.uint8)
        out = frame_io.encode_png_gpu(torch.from_numpy(img).cuda())
        _check_png_file(out.cpu().numpy().tobytes(), img)
    with pytest.raises(ValueError):
        frame_io.encode_png_gpu(torch.zeros(4, 4, 2, dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        frame_io.encode_png_gpu(torch.zeros(4, 4, 4, dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        frame_io.encode_png_gpu(torch.zeros(4, 4, 4, dtype=torch.uint8))


@pytest.mark.gpu
def test_gpu_frame_writer_leaves_the_same_pixels_and_the_same_npy_bytes(tmp_path):
    """GpuFrameWriter against write_frame_outputs (the host path, itself checked against the reference's formulas above) on real
    render() results: the three PNGs decode to the same pixels, the depth .npy is byte-identical to np.save's, frames in
    flight do not overwrite each other (12 frames through 3 slots), a second image size re-sizes the slots."""
    from autovfx_amd import renderer, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.gaussian_model import GaussianModel
    dev = "cuda:0"
    c = scenes.config_c2(P=30_000, seed=5).to(dev)
    model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3)
    bg = torch.zeros(3, device=dev)
    a, b = tmp_path / "host", tmp_path / "gpu"
    names = []
    with torch.no_grad(), frame_io.GpuFrameWriter(str(b), workers=2, slots=3) as w:
        for size in ((208, 120), (96, 64)):
            cams = orbit_cameras(12, *size)
            for i in range(12 if size[0] == 208 else 3):
                res = renderer.render(cams[i].to(dev), model, renderer.PipelineParams, bg)
                name = f"{size[0]}_{i:05d}"
                w.submit(name, res)
                frame_io.write_frame_outputs(str(a), name, res)
                names.append(name)
    for name in names:
        for sub, ext in (("images", ".png"), ("normal", ".png"), ("depth", ".png")):
            got = frame_io.decode_png((b / sub / (name + ext)).read_bytes())
            np.testing.assert_array_equal(got, frame_io.decode_png((a / sub / (name + ext)).read_bytes()), err_msg=f"{sub}/{name}")
        assert (b / "depth" / (name + ".npy")).read_bytes() == (a / "depth" / (name + ".npy")).read_bytes(), name
    img = frame_io.decode_png((b / "images" / (names[5] + ".png")).read_bytes())
    assert img.shape == (120, 208, 4) and img[..., 3].max() > 200


# ---- compressed file images (gsr_png_encode_deflate) ---------------------------------------------------------------------------
def _test_images(shape, seed):
    """What a frame can look like to the encoder: noise (incompressible: every block falls back to stored), a smooth ramp with a little
    noise (what a render is: small Paeth residuals), large flat areas (run-length matches), all of it in one image."""
    h, w, c = shape
    g = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    noise = g.integers(0, 256, shape).astype(np.uint8)
    ramp = ((xx[..., None] * (3 + np.arange(c)) + yy[..., None] * 2 + g.integers(0, 3, shape)) % 256).astype(np.uint8)
    flat = np.zeros(shape, np.uint8)
    flat[h // 3:, w // 4:] = ramp[h // 3:, w // 4:]
    mixed = ramp.copy()
    mixed[: h // 2, : w // 2] = noise[: h // 2, : w // 2]
    mixed[h // 2:, w // 2:] = 17
    return {"noise": noise, "ramp": ramp, "flat": flat, "mixed": mixed}


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1, 4), (1, 1, 3), (7, 5, 4), (64, 33, 3), (4, 4096, 4), (4, 4095, 4), (5, 3277, 3), (300, 100, 4),
                                   (540, 960, 4), (540, 960, 3), (1080, 1920, 4), (17, 1285, 3), (2, 21845, 3)])
@pytest.mark.parametrize("planar", [False, True])
def test_gpu_deflate_png_files_are_valid_and_decode_to_the_image(shape, planar):
    """gsr_png_encode_deflate: a compressed file comes off the GPU finished.  Checked like the stored files (chunk CRCs, zlib inflates
    the IDAT and verifies its Adler-32, PIL decodes the pixels) and, more strictly, the inflated stream must be the Paeth-filtered
    scanlines byte for byte.  Sizes around the 16 KB block (a stream of exactly one block, one byte less, rows that straddle blocks)."""
    h, w, c = shape
    for kind, img in _test_images(shape, h * 131 + w + c).items():
        src = torch.from_numpy(img).cuda()
        if planar:
            src = src.permute(2, 0, 1).contiguous()
        out = frame_io.encode_png_gpu_deflate(src, planar=planar)
        data = out.cpu().numpy().tobytes()
        assert len(data) <= frame_io.png_deflate_max_size(w, h, c)
        try:
            _check_png_file(data, img, paeth=True)
        except Exception as e:
            raise AssertionError(f"{kind} {shape} planar={planar}: {e!r}") from e
        if kind == "flat" and h * w >= 300 * 100:
            assert len(data) < 0.75 * img.size, (kind, len(data), img.size)
        if kind == "noise":                      # nothing to gain: stored blocks, a few bytes of framing per 16 KB
            assert len(data) <= img.size + h + 6 * (img.size // 16384 + 1) + 70


@pytest.mark.gpu
def test_gpu_deflate_png_constant_images_and_size_against_pil():
    PIL = pytest.importorskip("PIL.Image")
    for value in (0, 255):
        img = np.full((1080, 1920, 4), value, np.uint8)
        data = frame_io.encode_png_gpu_deflate(torch.from_numpy(img).cuda()).cpu().numpy().tobytes()
        _check_png_file(data, img, paeth=True)
        assert len(data) < img.size // 40
    # a rendered frame: within 1.5 x of what PIL (torchvision.utils.save_image's writer) makes of it at its default level
    from autovfx_amd import renderer, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.gaussian_model import GaussianModel
    c = scenes.config_c2(P=300_000, seed=5).to("cuda:0")
    model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3)
    with torch.no_grad():
        res = renderer.render(orbit_cameras(8, 960, 540)[3].to("cuda:0"), model, renderer.PipelineParams, torch.zeros(3, device="cuda:0"))
        rgba8, depth, normal = frame_io._frame_to_host(res)
    report = {}
    for name, img in (("rgba", rgba8), ("depth_preview", frame_io.depth2img(depth.squeeze(), 3.0)), ("normal", normal)):
        img = np.ascontiguousarray(img)
        data = frame_io.encode_png_gpu_deflate(torch.from_numpy(img).cuda()).cpu().numpy().tobytes()
        _check_png_file(data, img, paeth=True)
        b = io.BytesIO()
        PIL.fromarray(img).save(b, format="PNG")
        report[name] = (len(data), b.getbuffer().nbytes, img.size)
        assert len(data) <= 1.5 * b.getbuffer().nbytes, report
    print("deflate PNG bytes (ours, PIL, raw):", report)


@pytest.mark.gpu
@pytest.mark.parametrize("deflate", [True, False])
def test_gpu_frame_writer_in_both_png_modes(tmp_path, deflate):
    """GpuFrameWriter(deflate=...) against the host writer on real render() results, frames in flight through few slots."""
    from autovfx_amd import renderer, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.gaussian_model import GaussianModel
    PIL = pytest.importorskip("PIL.Image")
    dev = "cuda:0"
    c = scenes.config_c2(P=30_000, seed=6).to(dev)
    model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3)
    bg = torch.zeros(3, device=dev)
    a, b = tmp_path / "host", tmp_path / "gpu"
    cams = orbit_cameras(10, 320, 180)
    with torch.no_grad(), frame_io.GpuFrameWriter(str(b), workers=2, slots=3, deflate=deflate) as w:
        for i in range(10):
            res = renderer.render(cams[i].to(dev), model, renderer.PipelineParams, bg)
            w.submit(f"{i:05d}", res)
            frame_io.write_frame_outputs(str(a), f"{i:05d}", res)
    total = 0
    for i in range(10):
        for sub, ext in (("images", ".png"), ("normal", ".png"), ("depth", ".png")):
            got = np.asarray(PIL.open(b / sub / f"{i:05d}{ext}"))
            np.testing.assert_array_equal(got, np.asarray(PIL.open(a / sub / f"{i:05d}{ext}")), err_msg=f"{sub}/{i}")
            total += (b / sub / f"{i:05d}{ext}").stat().st_size
        assert (b / "depth" / f"{i:05d}.npy").read_bytes() == (a / "depth" / f"{i:05d}.npy").read_bytes()
    raw = 10 * 320 * 180 * 10
    assert (total < raw) if deflate else (total > raw)


@pytest.mark.gpu
def test_gpu_deflate_png_length_limits_a_skewed_code():
    """Token counts in powers of two make Huffman's tree a chain (also with the filter-type and end-of-block symbols mixed in): 18 literals
    -> code lengths up to 18 bits, which deflate does not allow.  The table kernel must limit the lengths to 15 and still hand zlib a COMPLETE code (an over- or under-subscribed one is rejected
    at the block header).  The image is built backwards from the residuals it should have (Paeth un-filtering, as a reader does)."""
    counts = [1 << k for k in range(18)]                                           # 1, 2, 4 ... 131 072: each symbol as frequent as all rarer ones together
    h, w, c = 63, 1387, 3                                                          # 63 x 1387 x 3 = 262 143 = their sum, exactly
    assert sum(counts) == h * w * c
    resid = np.concatenate([np.full(n, 3 + 7 * k, np.uint8) for k, n in enumerate(counts)])   # 18 distinct residual values
    np.random.default_rng(5).shuffle(resid)
    resid = resid.reshape(h, w * c)
    img = np.zeros((h, w, c), np.int32)
    for y in range(h):
        line = resid[y].reshape(w, c).astype(np.int32)
        up = img[y - 1] if y else np.zeros((w, c), np.int32)
        left, ul = np.zeros(c, np.int32), np.zeros(c, np.int32)
        for x in range(w):
            b = up[x]
            p = left + b - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - b), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, b, ul))
            left = (line[x] + pred) & 255
            img[y, x] = left
            ul = b
    img = img.astype(np.uint8)
    np.testing.assert_array_equal(_paeth_stream(img)[:, 1:], resid)                 # the residuals are what was asked for
    data = frame_io.encode_png_gpu_deflate(torch.from_numpy(img).cuda()).cpu().numpy().tobytes()
    _check_png_file(data, img, paeth=True)
    assert len(data) < img.size                                                     # (about two bits per byte)
