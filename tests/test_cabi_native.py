"""The C ABI used without Python: examples/render_raw.cpp (plain C++ / HIP, hipMalloc'ed scratch arenas, its own
stream) must produce exactly what the Python binding produces from the same inputs."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "examples", "bin", "render_raw")


def write_scene(path, cloud, cam, bg, W, H):
    f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype="<f4").tobytes()
    P, M = cloud.P, int(cloud.shs.shape[1])
    with open(path, "wb") as f:
        f.write(struct.pack("<6i", P, M, cloud.sh_degree, W, H, 0))
        f.write(struct.pack("<3f", cam.tanfovx, cam.tanfovy, 1.0))
        for t in (bg, cloud.means3D, cloud.shs, cloud.opacities, cloud.scales, cloud.rotations, cam.world_view_transform,
                  cam.full_proj_transform, cam.camera_center):
            f.write(f32(t))


def read_out(path, P, W, H):
    raw = open(path, "rb").read()
    n = struct.unpack_from("<i", raw, 0)[0]
    at = 4
    out = {}
    for k, count, dt in (("color", 3 * H * W, "<f4"), ("depth", H * W, "<f4"), ("alpha", H * W, "<f4"), ("radii", P, "<i4")):
        out[k] = np.frombuffer(raw, dtype=dt, count=count, offset=at)
        at += count * 4
    assert at == len(raw)
    return n, out


STREAM_BIN = os.path.join(ROOT, "examples", "bin", "render_stream")


@pytest.mark.parametrize("name", ["render_raw", "render_stream", "train_step_raw"])
def test_example_source_uses_only_the_public_header(name):
    src = open(os.path.join(ROOT, "examples", name + ".cpp")).read()
    includes = [l.split()[1] for l in src.splitlines() if l.startswith("#include")]
    assert '"../include/gsr.h"' in includes
    assert not any("torch" in i or "gsr_internal" in i or "Python" in i for i in includes)


@pytest.mark.gpu
@pytest.mark.parametrize("streams", [1, 3, 4])
def test_native_pipelined_frames_equal_the_one_shot_call(tmp_path, streams):
    """examples/render_stream.cpp keeps `streams` frames in flight from one host thread with gsr_forward_begin /
    gsr_forward_finish; its last frame must be byte-identical to render_raw's gsr_forward of the same scene."""
    assert os.path.exists(BIN) and os.path.exists(STREAM_BIN), "examples are not built: run __graft_entry__.build()"
    W, H = 400, 240
    cloud = scenes.config_c2(P=50_000, seed=17)
    cam = orbit_cameras(9, W, H)[5]
    scene = str(tmp_path / "scene.bin")
    write_scene(scene, cloud, cam, torch.tensor([0.0, 0.2, 0.4]), W, H)
    one, many = str(tmp_path / "one.bin"), str(tmp_path / "many.bin")
    r = subprocess.run([BIN, scene, one], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([STREAM_BIN, scene, many, "23", str(streams)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "frames/s" in r.stderr
    assert open(one, "rb").read() == open(many, "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("P,W,H", [(20_000, 320, 200), (0, 64, 48)])
def test_native_program_matches_python_binding(tmp_path, P, W, H):
    assert os.path.exists(BIN), "examples/bin/render_raw is not built: run __graft_entry__.build()"
    from helpers import run_hip
    cloud = scenes.config_c2(P=max(P, 1), seed=13)
    if P == 0:
        cloud = scenes.GaussianCloud(cloud.means3D[:0], cloud.opacities[:0], cloud.scales[:0], cloud.rotations[:0],
                                     cloud.shs[:0], None, cloud.sh_degree)
    cam = orbit_cameras(7, W, H)[3]
    bg = torch.tensor([0.1, 0.3, 0.2])
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    write_scene(scene, cloud, cam, bg, W, H)
    r = subprocess.run([BIN, scene, out, "3"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    n, got = read_out(out, P, W, H)
    if P == 0:
        assert n == 0 and not got["color"].any() and not got["alpha"].any()
        return
    want = run_hip(cloud, cam, bg=tuple(bg.tolist()))
    from diff_gaussian_rasterization import _C
    assert n == int(_C.last_layout()["counts"]["num_rendered"]) > 0      # num_rendered, the reference's return value
    for k in ("color", "depth", "alpha", "radii"):
        np.testing.assert_array_equal(got[k], np.asarray(want[k]).reshape(-1), err_msg=k)


TRAIN_BIN = os.path.join(ROOT, "examples", "bin", "train_step_raw")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["atomic", "deterministic"])
@pytest.mark.parametrize("M", [16, 4])
def test_native_raw_forward_and_backward_match_the_python_binding(tmp_path, M, mode):
    """examples/train_step_raw.cpp: gsr_forward_raw (a full call with the normal image) + gsr_backward_raw (colour and normal
    gradients given, depth / alpha NULL) from plain C++ against the Python binding on the same raw tensors: the four images
    and radii bit for bit; the seven gradient tensors bit for bit with GSR_OPT_BACKWARD_DETERMINISTIC on both sides (two
    processes, two allocators, same bits), and within 2e-4 of the array's scale in the default mode (sums of float atomics;
    a well-conditioned scene).  The program fills every output with 0xFF first: an element the library did not write would
    come back as NaN."""
    assert os.path.exists(TRAIN_BIN), "examples/bin/train_step_raw is not built: run __graft_entry__.build()"
    from diff_gaussian_rasterization import _C
    dev = "cuda:0"
    P, W, H, D = 15_000, 288, 176, {16: 3, 4: 1}[M]
    c = scenes.config_c1(P=P, seed=23)
    cam = orbit_cameras(7, W, H)[2]
    g = torch.Generator().manual_seed(4)
    raw = dict(xyz=c.means3D, ls=torch.log(c.scales), rot=c.rotations * (0.5 + 2 * torch.rand(P, 1, generator=g)),
               op=torch.logit(c.opacities.reshape(-1).clamp(1e-6, 1 - 1e-6)), dc=c.shs[:, :1].contiguous(), rest=c.shs[:, 1:M].contiguous())
    bg = torch.tensor([0.2, 0.1, 0.3])
    g_color = torch.randn(3, H, W, generator=g) / (H * W)
    g_normal = torch.randn(3, H, W, generator=g) / (H * W)
    model, out = str(tmp_path / "model.bin"), str(tmp_path / "out.bin")
    f32 = lambda t: np.ascontiguousarray(t.detach().cpu().numpy(), dtype="<f4").tobytes()
    with open(model, "wb") as f:
        f.write(struct.pack("<5i", P, M, D, W, H))
        f.write(struct.pack("<2f", cam.tanfovx, cam.tanfovy))
        for t in (bg, raw["xyz"], raw["ls"], raw["rot"], raw["op"], raw["dc"], raw["rest"], cam.world_view_transform,
                  cam.full_proj_transform, cam.camera_center, g_color, g_normal):
            f.write(f32(t))
    r = subprocess.run([TRAIN_BIN, model, out] + (["deterministic"] if mode == "deterministic" else []), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    blob = open(out, "rb").read()
    n = struct.unpack_from("<i", blob, 0)[0]
    at, got = 4, {}
    for k, count, dt in (("color", 3 * H * W, "<f4"), ("depth", H * W, "<f4"), ("alpha", H * W, "<f4"), ("normal", 3 * H * W, "<f4"),
                         ("radii", P, "<i4"), ("g_xyz", 3 * P, "<f4"), ("g_ls", 3 * P, "<f4"), ("g_rot", 4 * P, "<f4"), ("g_op", P, "<f4"),
                         ("g_dc", 3 * P, "<f4"), ("g_rest", 3 * P * (M - 1), "<f4"), ("g_2d", 3 * P, "<f4")):
        got[k] = np.frombuffer(blob, dtype=dt, count=count, offset=at)
        at += 4 * count
    assert at == len(blob)
    t = lambda a: a.to(dev).contiguous()
    camd = cam.to(dev)
    args = (t(bg), t(raw["xyz"]), t(raw["ls"]), t(raw["rot"]), t(raw["op"]), t(raw["dc"]), t(raw["rest"]))
    from autovfx_amd import _lib
    _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1 if mode == "deterministic" else 0)
    with torch.no_grad():
        fw = _C.rasterize_gaussians_raw(*args, 1.0, camd.world_view_transform, camd.full_proj_transform, camd.tanfovx, camd.tanfovy, H, W,
                                        D, camd.camera_center, False, False, want_normal=True, inference=False)
        bw = _C.rasterize_gaussians_raw_backward(*args, fw[4], 1.0, camd.world_view_transform, camd.full_proj_transform, camd.tanfovx,
                                                 camd.tanfovy, t(g_color), None, None, t(g_normal), D, camd.camera_center, fw[5], fw[0],
                                                 fw[6], fw[7], fw[3], False)
    torch.cuda.synchronize()
    _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)
    assert n == fw[0] > 0
    for k, want in (("color", fw[1]), ("depth", fw[2]), ("alpha", fw[3]), ("normal", fw[8]), ("radii", fw[4])):
        np.testing.assert_array_equal(got[k], want.cpu().numpy().reshape(-1), err_msg=k)
    for k, want in zip(("g_2d", "g_xyz", "g_ls", "g_rot", "g_op", "g_dc", "g_rest"), bw):
        a, b = got[k].astype(np.float64), want.cpu().numpy().reshape(-1).astype(np.float64)
        assert np.isfinite(a).all(), f"{k}: an element was not written"
        if mode == "deterministic":
            np.testing.assert_array_equal(got[k].view(np.uint32), want.cpu().numpy().reshape(-1).view(np.uint32), err_msg=k)
        else:
            assert np.abs(a - b).max() <= 2e-4 * np.abs(b).max() + 1e-6, k
    assert np.abs(got["g_rest"]).sum() > 0 if M > 1 else got["g_rest"].size == 0
