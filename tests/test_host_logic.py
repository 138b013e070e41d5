"""Host-side logic that needs no GPU: camera conventions, trajectory schema, the launch-syntax
rewriter of oracle/build_ref.py, and bench.py's algorithmic-bytes formula."""
import json
import math

import numpy as np
import torch

from autovfx_amd import cameras as cams


def test_projection_matrix_layout():
    """graphics_utils.py:52-72: P[0,0]=2n/(r-l), P[1,1]=2n/(t-b), P[3,2]=1, P[2,2]=f/(f-n), P[2,3]=-fn/(f-n)."""
    fx, fy = math.radians(60), math.radians(40)
    P = cams.projection_matrix(0.01, 100.0, fx, fy)
    assert abs(P[0, 0] - 1 / math.tan(fx / 2)) < 1e-6 and abs(P[1, 1] - 1 / math.tan(fy / 2)) < 1e-6
    assert P[3, 2] == 1 and abs(P[2, 2] - 100 / 99.99) < 1e-6 and abs(P[2, 3] + 1 / 99.99) < 1e-6
    assert P[0, 2] == 0 and P[1, 2] == 0


def test_camera_matrices_are_transposed_and_consistent():
    c2w = cams.orbit_c2w(4.0, 12, 30.0)[5]
    cam = cams.Camera.from_c2w(c2w, 500.0, 500.0, 640, 360)
    w2c = np.linalg.inv(c2w).astype(np.float32)
    np.testing.assert_allclose(cam.world_view_transform.numpy(), w2c.T, atol=1e-6)
    np.testing.assert_allclose(cam.camera_center.numpy(), c2w[:3, 3], atol=1e-5)
    full = (cam.projection_matrix.numpy().T @ w2c).T
    np.testing.assert_allclose(cam.full_proj_transform.numpy(), full, atol=1e-5)
    assert abs(cams.fov2focal(cam.FoVx, 640) - 500.0) < 1e-3
    # the orbit looks at the origin: it projects to the image centre, in front of the camera
    o = np.array([0, 0, 0, 1], np.float32) @ cam.full_proj_transform.numpy()
    assert abs(o[0] / o[3]) < 1e-5 and abs(o[1] / o[3]) < 1e-5 and o[3] > 0.2
    assert abs(np.linalg.norm(cam.camera_center.numpy()) - 4.0) < 1e-4


def test_trajectory_json_roundtrip(tmp_path):
    poses = cams.orbit_c2w(4.0, 7)
    d = cams.trajectory_dict("orbit", poses[::-1], 400.0, 410.0, 320.0, 180.0, 640, 360)
    assert [f["filename"] for f in d["frames"]][:2] == ["00000.png", "00001.png"] and d["camera_model"] == "OPENCV"
    p = tmp_path / "t.json"
    p.write_text(json.dumps(d))
    got = cams.cameras_from_trajectory(str(p), downscale_factor=2.0)
    assert len(got) == 7 and got[0].image_width == 320 and got[0].image_height == 180
    np.testing.assert_allclose(got[0].camera_center.numpy(), poses[-1][:3, 3], atol=1e-5)
    assert abs(cams.fov2focal(got[0].FoVx, 320) - 200.0) < 1e-3


def test_launch_rewriter():
    from oracle.build_ref import rewrite_launches
    src = "a();\nfoo<3> << <(P + 255) / 256, 256 >> > (\n  x, f(y, z));\nbar << <grid, block >> > (q)\nCHECK(, d)"
    out = rewrite_launches(src)
    assert "<<" not in out
    assert "gsr_shim::launch(dim3((P + 255) / 256), dim3(256), [&]() { foo<3>(\n  x, f(y, z)); })" in out
    assert "gsr_shim::launch(dim3(grid), dim3(block), [&]() { bar(q); })\nCHECK(, d)" in out


def test_algorithmic_bytes_worked_example():
    """C3 with V = 2.4 M, D = 15 M -> about 3.94 GB per frame, 6 sort passes."""
    import bench
    b = bench.algorithmic_bytes(3_000_000, 2_400_000, 15_000_000, 8160, 1920, 1080, 16)
    assert b["n_pass"] == 6
    assert abs(b["frame"] / 1e9 - 3.94) < 0.02
    assert b["blend"] == 44 * 15_000_000 + 24 * 1920 * 1080


def test_bench_finds_the_committed_pmc_traffic():
    """bench.py's roofline.traffic / roofline.frame.traffic come from the newest profiles/r*_traffic.json (written by
    scripts/pmc_reduce.py from separate --pmc passes, stamped with the commit they were taken at)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t = bench.pmc_traffic("c3")
    assert t is not None and "_traffic.json" in t["source"] and len(t["commit"]) >= 7
    blend = t["kernels"]["blend_quadrant_kernel"]
    assert 1e8 < blend["bytes_per_launch"] < 9e8 and 1.0 <= blend["launches_per_frame"] <= 8.0
    assert t["kernels"]["preprocess_kernel"]["bytes_per_launch"] > 1e8
    assert 1e9 < t["frame_bytes"] < 5e9
    assert bench.pmc_traffic("c2") is None and bench.pmc_traffic(None) is None   # only the C3 passes are kept


def test_slab_policy_is_host_logic():
    """Which inference calls are cut into depth slabs (gsr_plan_slabs; no device needed): the measured configurations
    land where the A/B runs put them, and the knobs do what include/gsr.h says."""
    from autovfx_amd import _lib
    T1080, T540 = 120 * 68, 60 * 34
    try:
        assert _lib.plan_slabs(8_900_000, 1920, 1080) == [400 * T1080]          # C3: 5.6 M pairs behind the first slab
        assert _lib.plan_slabs(2_310_000, 960, 540) == []                         # C2: 1.5 M behind it, below the pay-off
        assert _lib.plan_slabs(6_000_000, 960, 540) == [400 * T540]               # heavy cloud
        assert _lib.plan_slabs(17_700_000, 1920, 1080) == [400 * T1080]           # heavy at 1080p: still two slabs at most
        assert _lib.plan_slabs(0, 960, 540) == [] and _lib.plan_slabs(2 * 400 * T540 - 1, 960, 540) == []
        _lib.set_option(_lib.OPT_SLABS, 0)                                        # as many as the sizes call for: x3 each
        assert _lib.plan_slabs(17_700_000, 1920, 1080) == [400 * T1080, 1600 * T1080]
        _lib.set_option(_lib.OPT_SLABS, 1)
        assert _lib.plan_slabs(17_700_000, 1920, 1080) == []
        _lib.set_option(_lib.OPT_SLABS, 2)
        _lib.set_option(_lib.OPT_SLAB_MIN_REST, 0)
        assert _lib.plan_slabs(2_310_000, 960, 540) == [400 * T540]
        _lib.set_option(_lib.OPT_SLAB_FIRST, 6)
        assert _lib.plan_slabs(100 * T540, 960, 540) == [6 * T540]
        assert _lib.plan_slabs(10_000_000, 70_000, 70_000) == []                  # tile bit rows would not fit the LDS table
    finally:
        _lib.set_option(_lib.OPT_SLABS, 2)
        _lib.set_option(_lib.OPT_SLAB_FIRST, 400)
        _lib.set_option(_lib.OPT_SLAB_MIN_REST, 3_000_000)


def test_sugar_camera_mirrors_the_reference_call_site():
    """autovfx_amd.cameras.sugar_camera against a line-by-line transcription of sugar_model.py:2008-2032 with the reference's
    own getWorld2View / getProjectionMatrix formulas (graphics_utils.py:39-72) written out in numpy."""
    import math
    import numpy as np
    import torch
    from autovfx_amd.cameras import orbit_c2w, sugar_camera, orbit_cameras, sugar_orbit_cameras
    W, H, fovx = 320, 180, math.radians(60.0)
    fovy = 2 * math.atan(math.tan(fovx / 2) * H / W)
    gl = np.array(orbit_c2w(4.0, 8)[3])
    gl[:3, 1:3] *= -1                                   # a nerfstudio pose: OpenGL axes
    cx, cy, zn, zf = 0.09, -0.05, 0.01, 100.0
    cam = sugar_camera(gl[:3], fovx, fovy, W, H, cx, cy)
    # --- transcription of the call site ---
    c2w = np.concatenate((gl[:3].astype(np.float32), np.array([[0, 0, 0, 1]], np.float32)), 0)
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    R, T = np.transpose(w2c[:3, :3]), w2c[:3, 3]
    Rt = np.zeros((4, 4)); Rt[:3, :3] = R.transpose(); Rt[:3, 3] = T; Rt[3, 3] = 1.0
    wv = torch.Tensor(np.float32(Rt)).transpose(0, 1)
    tH, tW = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = tH * zn, tW * zn
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * zn / (right + right); P[1, 1] = 2.0 * zn / (top + top)
    P[0, 2] = 0.0; P[1, 2] = 0.0; P[3, 2] = 1.0; P[2, 2] = 1.0 * zf / (zf - zn); P[2, 3] = -(zf * zn) / (zf - zn)
    proj = P.transpose(0, 1).clone()
    proj[2, 0] = -cx; proj[2, 1] = -cy
    full = wv.unsqueeze(0).bmm(proj.unsqueeze(0)).squeeze(0)
    assert torch.equal(cam.world_view_transform, wv) and torch.equal(cam.full_proj_transform, full)
    assert torch.allclose(cam.camera_center, torch.tensor(gl[:3, 3], dtype=torch.float32))
    # a centred principal point is the vanilla GSCamera
    a, b = sugar_orbit_cameras(8, W, H, 0.0, 0.0)[3], orbit_cameras(8, W, H)[3]
    assert torch.allclose(a.full_proj_transform, b.full_proj_transform, atol=1e-6)
    assert cam.tanfovx == b.tanfovx
