"""Dynamic scenes (BASELINE configs[4]: a static scene plus rigidly moving inserted objects, re-composed every frame).

CPU: the numpy restatement (oracle/dynamic_oracle.py) against the REFERENCE's own ``transform_gaussians`` /
``merge_two_gaussians`` / ``matrix_to_quaternion`` / ``quaternion_multiply`` executed in PyTorch on the host.
GPU: ``gsr_place_object`` against the restatement (positions and rotations bit for bit, ``exp`` pinned to ``torch.exp`` on
the same device), ``DynamicScene`` frames against the reference-shaped composition in PyTorch on the GPU and -- rendered --
against the CPU oracle.
"""
import copy
import importlib.util
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

from autovfx_amd import gaussian_model as gm
from autovfx_amd import scenes
from oracle import dynamic_oracle as dyn

REF = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "gaussians_utils.py")), reason="reference tree not mounted")


def rot(axis, deg):
    a = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    th = math.radians(deg)
    return (np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K).astype(np.float32)


def models(P_base=4000, P_obj=700, seed=0):
    c = scenes.config_c2(P=P_base, seed=seed)
    base = gm.GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3)
    objs = {}
    for k, (name, s) in enumerate((("chair", 3), ("ball", 4))):
        o = scenes.config_c1(P=P_obj + 100 * k, seed=seed + s)
        m = gm.GaussianModel.from_activated(o.means3D * 0.3, o.opacities, o.scales * 0.5, o.rotations, o.shs, 3)
        g = torch.Generator().manual_seed(seed + 10 + k)
        m._rotation = m._rotation * (0.5 + torch.rand(m._rotation.shape[0], 1, generator=g))   # raw quaternions are not unit on disk
        objs[name] = (m, (0.05 * k, -0.02, 0.01))
    return base, objs


def raw(m):
    n = lambda t: t.detach().cpu().numpy()
    return {"xyz": n(m._xyz), "rotation": n(m._rotation), "log_scale": n(m._scaling), "opacity_raw": n(m._opacity),
            "features_dc": n(m._features_dc), "features_rest": n(m._features_rest)}


FRAMES = [
    [("chair", (0.5, 0.2, -0.1), rot((0, 0, 1), 25), 1.0)],
    [("chair", (0.6, 0.2, 0.3), rot((1, 2, 3), 140), 0.8), ("ball", (-0.4, 0.1, 0.2), rot((0, 1, 0), -75), 1.7)],
    [("ball", (-0.4, 0.1, 0.2), rot((1, 0, 0), 180), 2.5)],        # w = 0 quaternion: the argmax picks another candidate
    [],
    [("ball", (0.0, 0.0, 0.0), np.eye(3, dtype=np.float32), 1.0), ("chair", (0.1, 0.1, 0.1), rot((-1, 1, 0), 359), 0.3)],
]


def ulps(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia, ib = np.where(ia < 0, -(ia & 0x7FFFFFFF), ia), np.where(ib < 0, -(ib & 0x7FFFFFFF), ib)
    return int(np.abs(ia - ib).max()) if a.size else 0


def _reference_modules():
    """The reference's rotation_utils.py and gaussians_utils.py, loaded from where they lie with their heavy imports stubbed."""
    import ref_plyfile_stub as stub
    refgm = stub.reference_gaussian_model()
    for name in ("e3nn", "trimesh", "sugar", "sugar.gaussian_splatting", "sugar.gaussian_splatting.scene",
                 "sugar.gaussian_splatting.scene.gaussian_model"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["e3nn"].o3 = None
    sys.modules["sugar.gaussian_splatting.scene.gaussian_model"].GaussianModel = refgm.GaussianModel
    out = {}
    for name in ("rotation_utils", "gaussians_utils"):
        spec = importlib.util.spec_from_file_location("_reference_" + name, os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod     # gaussians_utils does `from rotation_utils import ...`
        spec.loader.exec_module(mod)
        out[name] = mod
    return refgm, out["rotation_utils"], out["gaussians_utils"]


@needs_reference
def test_restatement_against_the_reference_functions_on_the_host():
    refgm, ru, gu = _reference_modules()
    base, objs = models()

    def ref_model(m):
        r = refgm.GaussianModel(3)
        for k in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest"):
            setattr(r, k, getattr(m, k).clone())
        r.active_sh_degree = r.max_sh_degree   # as load_ply leaves it (gaussian_model.py:266)
        return r

    for fi, frame in enumerate(FRAMES):
        allg = copy.deepcopy(ref_model(base))
        for name, center, R, s in frame:   # scene_representation.py:357-372
            m, c0 = objs[name]
            # matrix_to_quaternion: bit for bit
            np.testing.assert_array_equal(ru.matrix_to_quaternion(torch.tensor(R)).numpy(), dyn.matrix_to_quaternion(R))
            tr = gu.transform_gaussians(ref_model(m), torch.Tensor(list(center)), torch.Tensor(R), s, torch.Tensor(list(c0)))
            x, q, ls = dyn.transform_raw(m._xyz.numpy(), m._rotation.numpy(), m._scaling.numpy(), center, R, s, c0)
            np.testing.assert_array_equal(tr._rotation.numpy(), q, err_msg=f"frame {fi} {name}: quaternion product")
            np.testing.assert_array_equal(tr._scaling.numpy(), ls, err_msg=f"frame {fi} {name}: log scales")
            # torch.matmul's summation order inside the BLAS call is its own: a few ulp of the LARGEST coordinate
            err = np.abs(tr._xyz.numpy().astype(np.float64) - x).max()
            assert err <= 4 * np.spacing(np.float32(np.abs(x).max())), (fi, name, err)
            allg = gu.merge_two_gaussians(allg, tr, 3)
        got = dyn.compose(raw(base), [(raw(objs[n][0]), c, R, s, objs[n][1]) for n, c, R, s in frame], base_sh_degree=3)
        assert got["means3D"].shape[0] == allg._xyz.shape[0]
        # the degree render() will use: the merged model is a fresh GaussianModel (0); an untouched deep copy keeps the scene's
        assert allg.active_sh_degree == (0 if frame else 3) == got["active_sh_degree"]
        np.testing.assert_array_equal(got["shs"], torch.cat((allg._features_dc, allg._features_rest), 1).numpy())
        assert ulps(got["opacities"], torch.sigmoid(allg._opacity).numpy()) <= 2
        assert ulps(got["scales"], torch.exp(allg._scaling).numpy()) <= 2   # (numpy and torch use different vectorised exp)
        np.testing.assert_allclose(got["rotations"], torch.nn.functional.normalize(allg._rotation).numpy(), rtol=0, atol=2e-7)


def test_placement_block_and_quaternion_mirror():
    from autovfx_amd import dynamic_scene as ds
    for name, center, R, s in (p for f in FRAMES for p in f):
        np.testing.assert_array_equal(ds.placement_block(center, R, s, (0.1, 0.2, 0.3)), dyn.placement_block(center, R, s, (0.1, 0.2, 0.3)))
        q = ds.matrix_to_quaternion(R)
        assert abs(float(np.linalg.norm(q)) - 1.0) < 1e-5 and q.dtype == np.float32


# ---------------------------------------------------------------- GPU ----------------------------------------------------------------
@pytest.mark.gpu
def test_place_object_against_the_restatement():
    from autovfx_amd.dynamic_scene import DynamicScene
    base, objs = models()
    scene = DynamicScene(base, objs)
    for fi, frame in enumerate(FRAMES):
        cloud = scene.compose(frame)
        torch.cuda.synchronize()
        want = dyn.compose(raw(base), [(raw(objs[n][0]), c, R, s, objs[n][1]) for n, c, R, s in frame])
        P = want["means3D"].shape[0]
        assert cloud.P == P == scene.P_base + sum(objs[n][0]._xyz.shape[0] for n, *_ in frame)
        nb = scene.P_base
        np.testing.assert_array_equal(cloud.means3D.cpu().numpy()[nb:], want["means3D"][nb:], err_msg=f"frame {fi}: positions")
        np.testing.assert_array_equal(cloud.rotations.cpu().numpy()[nb:], want["rotations"][nb:], err_msg=f"frame {fi}: rotations")
        np.testing.assert_array_equal(cloud.shs.cpu().numpy(), want["shs"])
        assert ulps(cloud.scales.cpu().numpy()[nb:], want["scales"][nb:]) <= 2   # (device expf against numpy's: last-bit differences)
        # the kernel's exp is the device library's, the one torch.exp uses on this GPU: bit for bit
        at = nb
        for n, c, R, s in frame:
            m = objs[n][0]
            ls = m._scaling.cuda() + np.float32(math.log(s))
            assert torch.equal(cloud.scales[at:at + ls.shape[0]], torch.exp(ls)), f"frame {fi} {n}: exp differs from torch.exp on the GPU"
            assert torch.equal(cloud.opacities[at:at + ls.shape[0]], torch.sigmoid(m._opacity.cuda()))
            at += ls.shape[0]
        # the base part is what the getters give, untouched by any frame
        assert torch.equal(cloud.scales[:nb], torch.exp(base._scaling.cuda())) and torch.equal(cloud.means3D[:nb], base._xyz.cuda())


@pytest.mark.gpu
def test_frames_match_the_reference_shaped_composition_and_the_cpu_oracle():
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.dynamic_scene import DynamicScene
    from oracle.dynamic_torch import reference_shaped_compose
    from autovfx_amd.frame_parallel import rasterize
    from oracle import cpu_oracle
    from test_parity_gpu import assert_images
    base, objs = models(P_base=30_000, P_obj=4000, seed=3)
    scene = DynamicScene(base, objs)
    dev = torch.device("cuda:0")
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    cams = orbit_cameras(len(FRAMES), 320, 180)
    for fi, frame in enumerate(FRAMES):
        cam = cams[fi].to(dev)
        with torch.no_grad():
            cloud = scene.compose(frame)
            assert cloud.sh_degree == (0 if frame else 3)   # the reference's merged model renders DC-only (module docstring)
            assert DynamicScene(base, objs, placed_sh_degree=None).compose(frame).sh_degree == 3
            color, depth, alpha, radii = [t.clone() for t in rasterize(cloud, cam, bg)]
            ref_cloud = reference_shaped_compose(base, objs, frame, dev)
            assert ref_cloud.sh_degree == cloud.sh_degree
            rc, rd, ra, rr = rasterize(ref_cloud, cam, bg)
        torch.cuda.synchronize()
        # same device, same torch kernels for exp / sigmoid / normalize; positions differ by the matmul's summation order
        assert cloud.P == ref_cloud.P
        assert float((cloud.means3D - ref_cloud.means3D).abs().max()) <= 1e-6 * float(ref_cloud.means3D.abs().max()) + 1e-7
        assert torch.equal(cloud.scales, ref_cloud.scales) and torch.equal(cloud.opacities, ref_cloud.opacities)
        assert torch.equal(cloud.rotations, ref_cloud.rotations)   # F.normalize's own summation order, bit for bit (round 4)
        got = {"color": color.cpu().numpy(), "depth": depth.cpu().numpy(), "alpha": alpha.cpu().numpy()}
        assert_images(f"dynamic_f{fi}_vs_torch", got, {"color": rc.cpu().numpy(), "depth": rd.cpu().numpy(), "alpha": ra.cpu().numpy()})
        assert int((radii != rr).sum()) <= max(1, cloud.P // 20000)
        # ... and the whole chain against the CPU: numpy composition -> C oracle
        w = dyn.compose(raw(base), [(raw(objs[n][0]), c, R, s, objs[n][1]) for n, c, R, s in frame])
        c_ = cams[fi]
        ref = cpu_oracle.forward(means3D=w["means3D"], opacities=w["opacities"], bg=np.array([0.1, 0.2, 0.3], np.float32), width=320,
                                 height=180, viewmatrix=c_.world_view_transform, projmatrix=c_.full_proj_transform,
                                 campos=c_.camera_center, tanfovx=c_.tanfovx, tanfovy=c_.tanfovy, sh_degree=w["active_sh_degree"], shs=w["shs"],
                                 scales=w["scales"], rotations=w["rotations"])
        assert_images(f"dynamic_f{fi}_vs_oracle", got, ref)
        assert int((radii.cpu().numpy() != ref["radii"]).sum()) <= max(1, cloud.P // 20000)


@pytest.mark.gpu
def test_render_of_a_composed_frame_through_the_model_getters():
    """``compose_model``: the frame as an object with the GaussianModel getters, so that ``render()`` (RGBA, depth, normal and
    pseudo-normal maps) runs on a moving scene.  The per-Gaussian minimum axis the kernel writes is the one
    ``get_minimum_axis`` computes from the composed scales / rotations, bit for bit; the maps agree with those of the
    reference-shaped composition."""
    from autovfx_amd import renderer
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.dynamic_scene import DynamicScene, FrameModel
    from oracle.dynamic_torch import reference_shaped_compose
    base, objs = models(P_base=20_000, P_obj=3000, seed=5)
    scene = DynamicScene(base, objs)
    dev = torch.device("cuda:0")
    bg = torch.tensor([0.0, 0.0, 0.0], device=dev)
    cam = orbit_cameras(6, 256, 144)[2].to(dev)
    for frame in (FRAMES[1], FRAMES[4], FRAMES[3]):
        with torch.no_grad():
            fm = scene.compose_model(frame)
            want_axis = gm.get_minimum_axis(fm.get_scaling, fm.get_rotation)
            assert torch.equal(fm.get_minimum_axis, want_axis), "minimum axis differs from get_minimum_axis on the composed tensors"
            out = renderer.render(cam, fm, renderer.PipelineParams, bg)
            got = {k: out[k].clone() for k in ("render", "depth", "normal", "pseudo_normal")}
            rc = reference_shaped_compose(base, objs, frame, dev)
            ref = renderer.render(cam, FrameModel(rc, gm.get_minimum_axis(rc.scales, rc.rotations).contiguous()), renderer.PipelineParams, bg)
        torch.cuda.synchronize()
        assert got["render"].shape == (4, 144, 256) and got["normal"].shape == (144, 256, 3)
        assert float((got["render"] - ref["render"]).abs().max()) <= 1e-4
        assert float((got["depth"] - ref["depth"]).abs().max()) <= 1e-4 * max(1.0, float(ref["depth"].abs().max()))
        # unit normals: compare where the pixel is covered (elsewhere both are the normalised background value)
        covered = ref["render"][3] > 0.5
        assert float((got["normal"] - ref["normal"])[covered].abs().max()) <= 2e-3


@pytest.mark.gpu
def test_an_object_placed_twice_needs_room_and_says_so():
    from autovfx_amd.dynamic_scene import DynamicScene
    base, objs = models(P_base=1000, P_obj=300)
    scene = DynamicScene(base, objs)
    two = [("ball", (0, 0, 0), np.eye(3), 1.0), ("ball", (1, 0, 0), np.eye(3), 1.0)]   # 2 x 400 <= 300 + 400 capacity? no
    with pytest.raises(ValueError, match="no room"):
        scene.compose(two + [("chair", (0, 1, 0), np.eye(3), 1.0)])
    assert scene.compose(two[:1]).P == 1000 + 400


# ---------------------------------------------------------------- masked subsets: the melting branch ----------------------------------------------------------------
def melting_frames(objs, seed=0):
    """Frames as the melting branch builds them (scene_representation.py:373-421): per object up to two melting meshes, each
    keeping the Gaussians whose closest mesh triangle survives -- here random masks of different densities, one empty, one full."""
    g = np.random.default_rng(seed)
    n = {k: int(m._xyz.shape[0]) for k, (m, _c) in objs.items()}
    mask = lambda k, p: g.random(n[k]) < p
    return [
        [("chair", None, None, None, mask("chair", 0.6))],
        [("chair", None, None, None, mask("chair", 0.3)), ("chair", None, None, None, mask("chair", 0.1)), ("ball", None, None, None, mask("ball", 0.9))],
        [("ball", None, None, None, np.zeros(n["ball"], bool)), ("chair", None, None, None, np.ones(n["chair"], bool))],
        [("ball", (0.2, 0.1, 0.0), rot((0, 1, 0), 30), 1.2, mask("ball", 0.5))],   # a subset that IS transformed
    ]


def test_masked_merge_restatement_matches_boolean_indexing():
    base, objs = models()
    for frame in melting_frames(objs):
        got = dyn.compose(raw(base), [(raw(objs[e[0]][0]), e[1], e[2], e[3], objs[e[0]][1], e[4]) for e in frame])
        want_xyz = [base._xyz.numpy()]
        for name, center, R, s, mask in frame:
            if center is None:
                want_xyz.append(objs[name][0]._xyz.numpy()[mask])
        if all(e[1] is None for e in frame):
            np.testing.assert_array_equal(got["means3D"], np.concatenate(want_xyz))
        assert got["means3D"].shape[0] == base._xyz.shape[0] + sum(int(np.asarray(e[4]).sum()) for e in frame)
        assert got["active_sh_degree"] == 0


@pytest.mark.gpu
def test_masked_subsets_are_merged_bit_for_bit_and_render_like_subset_then_merge():
    """``DynamicScene.compose`` with masks against the reference's sequence in PyTorch on the same GPU -- boolean-mask indexing
    of the six raw tensors, merge_two_gaussians, activation -- every composed tensor ``torch.equal`` (untransformed subsets are
    copied, never touched), and the rendered frame bit-identical."""
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.dynamic_scene import DynamicScene
    from autovfx_amd.frame_parallel import rasterize
    from oracle.dynamic_torch import reference_shaped_compose
    base, objs = models(P_base=30_000, P_obj=5000, seed=7)
    frames = melting_frames(objs, seed=3)
    cap_objs = {k: v for k, v in objs.items()}
    scene = DynamicScene(base, {**cap_objs, "chair2": objs["chair"]})   # room for the frame that merges two chair subsets
    frames = [[("chair2" if (i > 0 and e[0] == "chair" and any(p[0] == "chair" for p in fr[:i])) else e[0],) + tuple(e[1:])
               for i, e in enumerate(fr)] for fr in frames]
    objs_t = {**objs, "chair2": objs["chair"]}
    dev = torch.device("cuda:0")
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    cams = orbit_cameras(len(frames), 320, 180)
    for fi, frame in enumerate(frames):
        cam = cams[fi].to(dev)
        with torch.no_grad():
            cloud = scene.compose(frame)
            got = [t.clone() for t in rasterize(cloud, cam, bg)]
            ref_cloud = reference_shaped_compose(base, objs_t, frame, dev)
            want = rasterize(ref_cloud, cam, bg)
        torch.cuda.synchronize()
        assert cloud.P == ref_cloud.P == scene.P_base + sum(int(np.asarray(e[4]).sum()) for e in frame)
        assert cloud.sh_degree == ref_cloud.sh_degree == 0
        untransformed = all(e[1] is None for e in frame)
        for name in ("opacities", "scales", "rotations", "shs") + (("means3D",) if untransformed else ()):
            assert torch.equal(getattr(cloud, name), getattr(ref_cloud, name)), (fi, name)
        if untransformed:
            for a, b in zip(got, want):
                assert torch.equal(a, b), fi
        else:   # (positions of a transformed subset differ by the matmul's summation order, as for whole objects)
            assert float((cloud.means3D - ref_cloud.means3D).abs().max()) <= 1e-6 * float(ref_cloud.means3D.abs().max()) + 1e-7
            assert float((got[0] - want[0]).abs().max()) <= 1e-4
    # index lists and GPU masks are accepted too, and say so when they do not fit
    m = torch.zeros(objs["ball"][0]._xyz.shape[0], dtype=torch.bool, device=dev)
    m[::3] = True
    a = scene.compose([("ball", None, None, None, m)])
    b = scene.compose([("ball", None, None, None, torch.nonzero(m).reshape(-1).to(torch.int32))], slot=0)
    assert a.P == b.P == scene.P_base + int(m.sum())
    with pytest.raises(ValueError, match="entries"):
        scene.compose([("ball", None, None, None, np.ones(5, bool))])
    with pytest.raises(ValueError, match="untransformed"):
        scene.compose([("ball", None, np.eye(3), 1.0, None)])


@pytest.mark.gpu
def test_resident_subset_index_lists_are_validated_once():
    """an index list that already lives on the GPU used to reach gsr_place_object_subset unchecked -- an index >= n
    or a negative one is an out-of-bounds device read.  It is checked now the first time a given list is seen (one host read of
    its min / max) and remembered by (address, length, version): a bad list raises, a good one is not read back again, and an
    in-place edit is checked anew."""
    from autovfx_amd.dynamic_scene import DynamicScene
    base, objs = models()
    scene = DynamicScene(base, objs)
    name = next(iter(objs))
    n = int(objs[name][0]._xyz.shape[0])
    good = torch.arange(0, n, 3, dtype=torch.int32, device="cuda")
    scene.compose([(name, None, None, None, good)])
    seen = dict(scene._checked_subsets)
    assert len(seen) == 1                                      # (keyed by the caller's tensor: address, length, version, object size)
    scene.compose([(name, None, None, None, good)])
    assert scene._checked_subsets == seen                      # the second frame did not look again
    for bad_value in (n, -1):
        bad = good.clone()
        bad[5] = bad_value
        with pytest.raises(ValueError, match="out of range"):
            scene.compose([(name, None, None, None, bad)])
    good[7] = n + 3                                            # an in-place edit bumps the version: checked again
    with pytest.raises(ValueError, match="out of range"):
        scene.compose([(name, None, None, None, good)])
