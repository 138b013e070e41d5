"""GPU parity of the backward pass: ``loss.backward()`` through the drop-in API (-> gsr_backward -> gfx950
kernels) against the fp64 gradient truth, the CPU oracle's fp32 backward and golden vectors from the reference's backward.cu.

Bar (tests/helpers.py: assert_gradients_vs_truth): per gradient array, in the max norm,
    |hip - truth| <= max(2e-4 * max|truth| + 1e-6, 4 * |reference_fp32 - truth|)
with truth = backward.cu's formulas evaluated in double on the fp32 forward state (oracle/gsr_oracle.c:
gsro_backward_f64) and reference_fp32 = the CPU oracle, which is backward.cu bit for bit.  Exact bit equality with the
reference is not attainable for sums formed with atomics (neither here nor in the reference), and for an ill-conditioned
Gaussian two fp32 samples differ by far more than the last bits -- the truth says how far either is allowed to be.
Every case runs in both modes of the library: float atomics (default) and GSR_OPT_BACKWARD_DETERMINISTIC, whose results
are a pure function of the inputs (asserted: same bits twice).
"""
import os

import numpy as np
import pytest
import torch

from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
from autovfx_amd.scenes import GaussianCloud
from oracle import cpu_oracle

from helpers import assert_gradients_vs_truth, decode_scratch, dump_on_failure, oracle_kwargs, settings_for
from test_oracle_backward import GOLDEN_BW, cpu_cov3d, load_bw_case, pixel_grads
from test_parity_gpu import report

pytestmark = pytest.mark.gpu


# atomic / deterministic: the forward is a FULL call (one list per tile; its scratch is decoded and compared with the oracle's lists);
# slabs: the forward is an inference call cut into as many depth slabs as the scene gives (GSR_OPT_GRAD_SLABS, what the binding
# does by default in grad mode -- here with slabs small enough that even these scenes are cut up) and the backward walks the
# slabs' segments; float atomics.
MODES = ("atomic", "deterministic", "slabs")


def hip_backward(cloud, cam, pg, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, sh_degree=None, cov3D_precomp=None,
                 device="cuda:0", cull=True, tanfov=None, mode="atomic"):
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import GaussianRasterizer
    c = cloud.to(device)
    st = settings_for(cam, device, bg, scale_modifier, cloud.sh_degree if sh_degree is None else sh_degree)
    if tanfov is not None:
        st = st._replace(tanfovx=tanfov[0], tanfovy=tanfov[1])
    leaf = lambda t: None if t is None else t.clone().requires_grad_(True)
    means3D, opac = leaf(c.means3D), leaf(c.opacities)
    shs, colors = leaf(c.shs), leaf(c.colors_precomp)
    means2D = torch.zeros_like(c.means3D, requires_grad=True)
    kw, cov = {}, None
    if cov3D_precomp is not None:
        cov = torch.as_tensor(cov3D_precomp, dtype=torch.float32, device=device).clone().requires_grad_(True)
        kw["cov3D_precomp"] = cov
        scales = rots = None
    else:
        scales, rots = leaf(c.scales), leaf(c.rotations)
        kw["scales"], kw["rotations"] = scales, rots
    _lib.set_option(_lib.OPT_TILE_CULL, 1 if cull else 0)
    _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1 if mode == "deterministic" else 0)
    _lib.set_option(_lib.OPT_GRAD_SLABS, 1 if mode == "slabs" else 0)
    if mode == "slabs":
        _lib.set_option(_lib.OPT_SLABS, 0)
        _lib.set_option(_lib.OPT_SLAB_FIRST, 6)
        _lib.set_option(_lib.OPT_SLAB_MIN_REST, 0)
    try:
        color, depth, alpha, radii = GaussianRasterizer(st)(means3D=means3D, means2D=means2D, opacities=opac, shs=shs,
                                                            colors_precomp=colors, **kw)
        # the forward's own scratch (what the backward is about to read), decoded BEFORE the backward runs: a wrong
        # gradient can then be traced to a wrong list / order / last contributor instead of being one number too far off
        from diff_gaussian_rasterization import _C
        torch.cuda.synchronize()
        saved = color.grad_fn.saved_tensors   # (.., radii, sh, geom, binning, image, alpha): __init__.py save_for_backward
        if mode == "slabs":
            fwd = {"slab_pairs": _C.last_layout()["slab_pairs"]} if cloud.P else {}
        else:
            fwd = decode_scratch(_C.last_layout(), saved[7], saved[8], saved[9], cloud.P, cam.image_width, cam.image_height) if cloud.P else {}
        t = lambda a: torch.from_numpy(a).to(device)
        loss = (color * t(pg["dL_dcolor"])).sum() + (depth * t(pg["dL_ddepth"])).sum() + (alpha * t(pg["dL_dalpha"])).sum()
        loss.backward()
        torch.cuda.synchronize()
    finally:
        _lib.set_option(_lib.OPT_TILE_CULL, 1)
        _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)
        _lib.set_option(_lib.OPT_GRAD_SLABS, 1)
        _lib.set_option(_lib.OPT_SLABS, 2)
        _lib.set_option(_lib.OPT_SLAB_FIRST, 400)
        _lib.set_option(_lib.OPT_SLAB_MIN_REST, 3000000)
    g = lambda x: None if x is None or x.grad is None else x.grad.cpu().numpy()
    return {"color": color.detach().cpu().numpy(), "depth": depth.detach().cpu().numpy(), "alpha": alpha.detach().cpu().numpy(),
            "dL_dmeans3D": g(means3D), "dL_dmeans2D": g(means2D),
            "dL_dopacity": g(opac), "dL_dsh": g(shs), "dL_dcolors": g(colors), "dL_dscales": g(scales),
            "dL_drotations": g(rots), "dL_dcov3D": g(cov), "radii": radii.cpu().numpy(), "fwd": fwd, "cull": cull, "mode": mode}


def assert_forward_state(name, hip, fref):
    """The forward half of a backward case, against the oracle's forward with intermediates: integers bit-exact
    (radii, depth order; with tile culling off also the lists, their ranges and every pixel's last contributor -- with
    it on the lists are thinner, so they are checked for order and against the unculled run's images), images within
    the forward tolerance.  Runs BEFORE the gradient comparison so that a failure says which stage went wrong."""
    from test_parity_gpu import RGB_TOL, FLIP_PPM
    f = hip["fwd"]
    np.testing.assert_array_equal(hip["radii"], fref["radii"], err_msg=f"{name}: radii")
    vis = fref["radii"] > 0
    V = int(vis.sum())
    ids = np.nonzero(vis)[0]
    expect = ids[np.lexsort((ids, fref["depths"][ids].view(np.uint32)))]
    if hip.get("mode") == "slabs":
        pass   # (an inference call's scratch: its lists are compared with the full call's at every parity case of test_parity_gpu.py)
    elif not hip["cull"]:
        np.testing.assert_array_equal(f["depth_order"][:V], expect.astype(np.uint32), err_msg=f"{name}: depth order")
        np.testing.assert_array_equal(f["point_list"], fref["point_list"], err_msg=f"{name}: point_list")
        np.testing.assert_array_equal(f["ranges"], fref["ranges"], err_msg=f"{name}: ranges")
        bad = int((f["n_contrib"] != fref["n_contrib"]).sum())
        assert bad <= int(np.ceil(FLIP_PPM * 1e-6 * f["n_contrib"].size)), f"{name}: n_contrib differs on {bad} pixels"
    else:
        # culled: splats whose every tile is dead leave the depth order; what is left keeps the oracle's order, and each
        # tile's list is ascending in (depth bits, id) -- the property the blend and the backward rely on
        steps = np.diff(f["point_offsets"].astype(np.int64), prepend=0)
        kept = f["depth_order"][steps > 0]   # (the order behind the visible Gaussians is undefined: GSR_OPT_DEPTH_DROP)
        pos = np.full(fref["radii"].shape[0], -1, np.int64)
        pos[expect] = np.arange(V)
        assert (pos[kept] >= 0).all() and (np.diff(pos[kept]) > 0).all(), f"{name}: culled depth order is not a subsequence"
        pl, tk = f["point_list"], f["tile_keys"]
        keys = (fref["depths"][pl].view(np.uint32).astype(np.uint64) << np.uint64(32)) | pl.astype(np.uint64)
        if pl.size > 1:
            same_tile = tk[1:] == tk[:-1]
            assert (tk[1:] >= tk[:-1]).all(), f"{name}: tile keys not ascending"
            out_of_order = int((same_tile & ~(keys[1:] > keys[:-1])).sum())
            assert out_of_order == 0, f"{name}: {out_of_order} list entries out of (depth, id) order inside their tile"
    dmax = max(1.0, float(np.abs(fref["depth"]).max()))
    npx = fref["alpha"].size
    for key, tol in (("color", RGB_TOL), ("alpha", RGB_TOL), ("depth", RGB_TOL * dmax)):
        err = np.abs(hip[key].astype(np.float64) - fref[key].astype(np.float64))
        bad = int((err > tol).sum())
        assert bad <= int(np.ceil(FLIP_PPM * 1e-6 * npx)), f"{name}: forward {key} off on {bad} px (max {err.max():.3e})"


_ORACLE_CACHE = {}


def oracles(name, kw, yardstick=False):
    """(reference fp32 backward, fp64 truth, fp32 noise yardstick or None) of one case, computed once for both modes of the library."""
    if name not in _ORACLE_CACHE:
        _ORACLE_CACHE.clear()   # one case at a time: a full-size case holds a gigabyte
        truth = cpu_oracle.backward_f64(**kw)
        _ORACLE_CACHE[name] = (cpu_oracle.backward(**kw), truth, cpu_oracle.fp32_noise(truth, **kw) if yardstick else None)
    return _ORACLE_CACHE[name]


def check_case(name, cloud, cam, pg, keys, mode, hip_kw=None, yardstick=False, **okw):
    """One backward case in one mode of the library: forward state first (integers bit-exact against the oracle), then every
    gradient against the truth; on any failure the inputs, the HIP results (with the decoded forward scratch) and the
    oracle's are dumped to gpurun_out/failures/.  In deterministic mode a second run must give the same bits."""
    kw = oracle_kwargs(cloud, cam, **okw)
    kw.update(pg)
    ref, truth, noise = oracles(name, kw, yardstick)
    hkw = dict(bg=tuple(float(v) for v in okw.get("bg", (0.0, 0.0, 0.0))), scale_modifier=okw.get("scale_modifier", 1.0),
               sh_degree=okw.get("sh_degree"), cov3D_precomp=okw.get("cov3D_precomp"))
    hkw.update(hip_kw or {})
    hip = hip_backward(cloud, cam, pg, mode=mode, **hkw)
    fref = cpu_oracle.forward(intermediates=True, **oracle_kwargs(cloud, cam, **okw))
    with dump_on_failure(f"bw_{name}_{mode}", hip={k: v for k, v in hip.items() if k != "fwd"}, fwd=hip["fwd"], ref=ref, fref=fref, pg=pg,
                         cloud={"means3D": cloud.means3D, "opacities": cloud.opacities, "scales": cloud.scales,
                                "rotations": cloud.rotations}):
        assert_forward_state(name, hip, fref)
        assert_gradients_vs_truth(f"{name}:{mode}", hip, ref, truth, keys, noise=noise)
        if mode == "deterministic":
            again = hip_backward(cloud, cam, pg, mode=mode, **hkw)
            assert_same_bits(name, hip, again, keys)
    return hip, ref


def assert_same_bits(name, a, b, keys):
    for k in keys:
        if a.get(k) is None:
            continue
        x, y = np.ascontiguousarray(a[k]).view(np.uint32), np.ascontiguousarray(b[k]).view(np.uint32)
        assert np.array_equal(x, y), f"{name}: {k} differs between two deterministic runs in {int((x != y).sum())} elements"


KEYS_SH = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations")
KEYS_PRE = ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dscales", "dL_drotations")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("cull", [True, False])
def test_backward_sh_scene(cull, mode):
    cloud, cam = scenes.config_c1(P=6000, seed=31), scenes.c1_camera(192, 128)
    check_case("sh", cloud, cam, pixel_grads(cam, 5), KEYS_SH, mode, hip_kw={"cull": cull}, bg=(0.1, 0.2, 0.3))


@pytest.mark.parametrize("mode", MODES)
def test_backward_precomputed_colours_and_orbit_camera(mode):
    cloud, cam = scenes.config_c4(P=15000, seed=32), orbit_cameras(8, 240, 135)[3]
    check_case("precomp", cloud, cam, pixel_grads(cam, 6), KEYS_PRE, mode, bg=(1.0, 1.0, 1.0))


@pytest.mark.parametrize("mode", MODES)
def test_backward_cov3d_precomp_scale_modifier_low_degree(mode):
    cloud, cam = scenes.config_c1(P=2500, seed=33), scenes.c1_camera(100, 70)
    pg = pixel_grads(cam, 7)
    cov = cpu_cov3d(cloud)
    check_case("cov3d_deg1", cloud, cam, pg, ("dL_dmeans3D", "dL_dmeans2D", "dL_dopacity", "dL_dsh", "dL_dcov3D"), mode,
               cov3D_precomp=cov, sh_degree=1, bg=(0.5, 0.5, 0.5))
    cloud2 = scenes.config_c1(P=2500, seed=34)
    check_case("scale_mod", cloud2, cam, pg, KEYS_SH, mode, scale_modifier=1.7)


@pytest.mark.parametrize("mode", MODES)
def test_backward_big_splats_and_ragged_image(mode):
    cloud, cam = scenes.config_c1(P=500, seed=35), scenes.c1_camera(131, 77)
    cloud.scales[:40] *= 20.0
    check_case("big_ragged", cloud, cam, pixel_grads(cam, 8), KEYS_SH, mode)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("path", GOLDEN_BW, ids=[os.path.basename(p)[:-4] for p in GOLDEN_BW])
def test_backward_matches_reference_golden_vectors(path, mode):
    """The committed golden vectors are backward.cu's own fp32 results (tests/golden/make_golden.py from oracle/_ref): they are
    the ``ref32`` of the bar, the truth is computed from the same inputs."""
    from autovfx_amd.cameras import Camera
    kw, ref = load_bw_case(path)
    t = lambda k: None if k not in kw else torch.from_numpy(np.asarray(kw[k]))
    cloud = GaussianCloud(t("means3D"), t("opacities"), t("scales"), t("rotations"), t("shs"), t("colors_precomp"),
                          kw["sh_degree"])
    cam = Camera(kw["width"], kw["height"], 2 * np.arctan(kw["tanfovx"]), 2 * np.arctan(kw["tanfovy"]),
                 t("viewmatrix"), t("projmatrix"), t("projmatrix"), t("campos"))
    pg = {k: kw[k] for k in ("dL_dcolor", "dL_ddepth", "dL_dalpha")}
    hip = hip_backward(cloud, cam, pg, bg=tuple(float(v) for v in kw["bg"]), scale_modifier=kw["scale_modifier"],
                       tanfov=(kw["tanfovx"], kw["tanfovy"]), mode=mode)
    truth = cpu_oracle.backward_f64(**kw)
    assert_gradients_vs_truth(f"golden:{os.path.basename(path)[:-4]}:{mode}", hip, ref, truth, KEYS_SH if cloud.shs is not None else KEYS_PRE)


def test_training_step_reduces_loss():
    """The call shape of the reference's training loops (train.py:84-134, scene_representation.py:495-520):
    render, L1 to a target, backward, Adam step -- the loss must go down."""
    dev = "cuda:0"
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = scenes.c1_camera(96, 96)
    target_cloud = scenes.config_c1(P=3000, seed=40).to(dev)
    st = settings_for(cam, dev, sh_degree=3)
    with torch.no_grad():
        target = GaussianRasterizer(st)(target_cloud.means3D, None, target_cloud.opacities, shs=target_cloud.shs,
                                        scales=target_cloud.scales, rotations=target_cloud.rotations)[0]
    shs = (target_cloud.shs + 0.3 * torch.randn_like(target_cloud.shs)).requires_grad_(True)
    logit_o = torch.logit(target_cloud.opacities.clamp(0.02, 0.98)).add(0.5).requires_grad_(True)
    opt = torch.optim.Adam([shs, logit_o], lr=0.02)
    losses = []
    for _ in range(12):
        screen = torch.zeros_like(target_cloud.means3D, requires_grad=True)
        img = GaussianRasterizer(st)(target_cloud.means3D, screen, torch.sigmoid(logit_o), shs=shs,
                                     scales=target_cloud.scales, rotations=target_cloud.rotations)[0]
        loss = (img - target).abs().mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    report("bw:train", first=losses[0], last=losses[-1])
    assert losses[-1] < 0.7 * losses[0], losses


@pytest.mark.parametrize("seed", range(10))
def test_backward_randomised_configurations(seed):
    mode = MODES[(seed // 2) % 2]   # seeds 0 1 4 5 8 9 atomic, 2 3 6 7 deterministic (culling alternates with the seed)
    rng = np.random.default_rng(500 + seed)
    P = int(rng.choice([1, 3, 64, 65, 400, 2000]))
    W, H = int(rng.integers(1, 200)), int(rng.integers(1, 150))
    cloud = scenes.config_c1(P=P, seed=600 + seed)
    if seed % 3 == 1 and P >= 8:
        cloud.scales[: P // 8] *= 25.0
    if seed % 3 == 2:
        cloud.opacities *= 0.05
    deg = int(rng.integers(0, 4))
    bg = tuple(float(v) for v in rng.uniform(0, 1, 3))
    cam = scenes.c1_camera(W, H, fovx_deg=float(rng.uniform(35, 95)))
    pg = pixel_grads(cam, seed)
    check_case(f"rand{seed}", cloud, cam, pg, KEYS_SH, mode, hip_kw={"cull": bool(seed % 2)}, bg=bg, sh_degree=deg)


@pytest.mark.parametrize("mode", MODES)
def test_backward_c2_full_size_vs_oracle(mode):
    """BASELINE configs[1] stand-in at full size (1 M Gaussians, 960x540, orbit frame 100): every gradient against the
    truth and the CPU oracle's backward, same bar as the small cases."""
    cloud, cam = scenes.config_c2(), orbit_cameras(200, 960, 540)[100]
    check_case("c2_full_1M", cloud, cam, pixel_grads(cam, 5), KEYS_SH, mode)


def test_rendered_gaussians_behind_an_opaque_front_get_exact_zeros_like_the_oracle():
    """The per-Gaussian backward kernel does not read the parameters of a RENDERED Gaussian (radii > 0) whose ten sums are all
    zero -- nobody composited it -- and writes zeros (gsr_backward.hip: preprocess_backward_lane).  A wall of opaque splats in
    front of the cloud leaves most of the cloud rendered but untouched: the set of all-zero gradient rows must be the oracle's
    (backward.cu multiplies by the zero sums), and the touched rows keep the usual bar."""
    g = torch.Generator().manual_seed(12)
    cloud, cam = scenes.config_c1(P=5000, seed=41), scenes.c1_camera(160, 128)   # camera at (0, 0, -4) looking down +z
    # a 12 x 10 wall of big, opaque, isotropic splats one unit in front of the camera, covering the whole image
    xs, ys = torch.meshgrid(torch.linspace(-0.8, 0.8, 12), torch.linspace(-0.7, 0.7, 10), indexing="xy")
    wall = torch.stack([xs.reshape(-1), ys.reshape(-1), torch.full((xs.numel(),), -3.0)], 1)
    W = wall.shape[0]
    rot = torch.zeros(W, 4); rot[:, 0] = 1.0
    both = GaussianCloud(means3D=torch.cat([wall, cloud.means3D]), opacities=torch.cat([torch.full((W, 1), 0.999), cloud.opacities.reshape(-1, 1)]),
                         scales=torch.cat([torch.full((W, 3), 0.15), cloud.scales]), rotations=torch.cat([rot, cloud.rotations]),
                         shs=torch.cat([torch.randn((W,) + tuple(cloud.shs.shape[1:]), generator=g) * 0.3, cloud.shs]))
    hip, ref = check_case("opaque_wall", both, cam, pixel_grads(cam, 9), KEYS_SH, "atomic")
    rendered = hip["radii"] > 0
    rows = lambda d: np.concatenate([np.asarray(d[k]).reshape(both.P, -1) for k in KEYS_SH], 1)
    idle_ref, idle_hip = (rows(ref) == 0).all(1), (rows(hip) == 0).all(1)
    assert int((rendered & idle_ref).sum()) > both.P // 2, "the wall does not hide the cloud: the case tests nothing"
    # (a composited-or-not decision at the 1/255 and 1e-4 thresholds may flip between the two for a pixel or two)
    assert int((idle_hip != idle_ref).sum()) <= 3, np.nonzero(idle_hip != idle_ref)[0][:10]


def test_colour_only_loss_skips_the_depth_and_alpha_terms_with_the_same_gradients():
    """A loss that reads only the colour image (the reference's training loops: train.py:84-134,
    scene_representation.py:495-520) leaves depth and alpha without a gradient: autograd then hands None to the backward
    (set_materialize_grads(False)), gsr_backward gets NULL for both images and the per-pixel pass runs without their terms.
    Gradients must equal those of the same loss with explicit zero weights on depth and alpha (which takes the full kernel)
    -- to the tolerance of sums formed with atomics -- and the oracle's."""
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = "cuda:0"
    cloud, cam = scenes.config_c1(P=20_000, seed=61), orbit_cameras(8, 320, 200)[5]
    pg = pixel_grads(cam, 12)
    pg["dL_ddepth"] = np.zeros_like(pg["dL_ddepth"])
    pg["dL_dalpha"] = np.zeros_like(pg["dL_dalpha"])
    kw = oracle_kwargs(cloud, cam, bg=(0.2, 0.1, 0.3))
    kw.update(pg)
    ref, truth = cpu_oracle.backward(**kw), cpu_oracle.backward_f64(**kw)
    full = hip_backward(cloud, cam, pg, bg=(0.2, 0.1, 0.3))            # zero-filled depth / alpha gradients: every term runs
    c = cloud.to(dev)
    st = settings_for(cam, dev, (0.2, 0.1, 0.3), 1.0, cloud.sh_degree)
    leaves = [t.clone().requires_grad_(True) for t in (c.means3D, c.opacities, c.shs, c.scales, c.rotations)]
    means2D = torch.zeros_like(c.means3D, requires_grad=True)
    color, depth, alpha, radii = GaussianRasterizer(st)(means3D=leaves[0], means2D=means2D, opacities=leaves[1], shs=leaves[2],
                                                        scales=leaves[3], rotations=leaves[4])
    (color * torch.from_numpy(pg["dL_dcolor"]).to(dev)).sum().backward()   # depth and alpha never enter the loss
    torch.cuda.synchronize()
    lean = {"dL_dmeans3D": leaves[0].grad, "dL_dmeans2D": means2D.grad, "dL_dopacity": leaves[1].grad, "dL_dsh": leaves[2].grad,
            "dL_dscales": leaves[3].grad, "dL_drotations": leaves[4].grad}
    lean = {k: v.cpu().numpy() for k, v in lean.items()}
    assert_gradients_vs_truth("colour_only", lean, ref, truth, KEYS_SH)
    assert_gradients_vs_truth("colour_only_full_kernel", full, ref, truth, KEYS_SH)
    # a loss that touches nothing of the rasterizer's outputs but its depth: colour arrives as None and is zero-filled
    for t in leaves:
        t.grad = None
    color, depth, alpha, radii = GaussianRasterizer(st)(means3D=leaves[0], means2D=torch.zeros_like(c.means3D), opacities=leaves[1],
                                                        shs=leaves[2], scales=leaves[3], rotations=leaves[4])
    depth.sum().backward()
    assert leaves[0].grad is not None and torch.isfinite(leaves[0].grad).all() and float(leaves[0].grad.abs().sum()) > 0
    assert float(leaves[2].grad.abs().sum()) == 0.0   # no colour gradient -> none for the SH coefficients


@pytest.mark.parametrize("mode", MODES)
def test_backward_c4_full_size_vs_oracle(mode):
    """BASELINE configs[3] at bench size (200 k flat SuGaR-style Gaussians, colors_precomp, 960x540, SuGaR's off-centre
    principal-point camera, orbit frame 25): every gradient against the truth.  These are the ill-conditioned ones (thin axis
    3e-4 against 2e-2 in plane): the reference's own fp32 backward is 1.7e-3 of the scale away from the truth in dL_dscales,
    and exact per-Gaussian sums rounded to fp32 move its result by 1e-3 (profiles/r05_gradient_truth.md) -- the round-4 bar
    of 2e-4 between two fp32 samples was inside that noise.  Ill-conditioned by design: the bar also takes the fp32 noise
    yardstick (cpu_oracle.fp32_noise)."""
    from autovfx_amd.cameras import sugar_orbit_cameras
    cloud, cam = scenes.config_c4(), sugar_orbit_cameras(50, 960, 540)[25]
    assert cloud.P == 200_000
    check_case("c4_full_200k", cloud, cam, pixel_grads(cam, 8), KEYS_PRE, mode, yardstick=True, bg=(1.0, 1.0, 1.0))


def test_backward_sh_degree_four_with_25_coefficients_leaves_higher_bands_zero():
    """A vanilla GaussianModel(sh_degree=4) carries M = 25 coefficients; computeColorFromSH (forward.cu:20-71 and its
    backward, backward.cu:20-138) only knows bands 0..3, so coefficients 16..24 receive exactly zero gradient (the
    reference's binding zero-fills dL_dsh, rasterize_points.cu:158-168) and the first 16 equal the M = 16 result."""
    cloud, cam = scenes.config_c1(P=3000, seed=41), scenes.c1_camera(128, 96)
    g = torch.Generator().manual_seed(2)
    wide = GaussianCloud(cloud.means3D, cloud.opacities, cloud.scales, cloud.rotations,
                         torch.cat((cloud.shs, torch.randn(cloud.P, 9, 3, generator=g)), 1).contiguous(), None, 4)
    pg = pixel_grads(cam, 9)
    a = hip_backward(cloud, cam, pg, sh_degree=3, mode="deterministic")   # fixed summation order: the two runs can be compared bit for bit
    b = hip_backward(wide, cam, pg, sh_degree=4, mode="deterministic")
    assert b["dL_dsh"].shape == (cloud.P, 25, 3)
    assert not b["dL_dsh"][:, 16:].any(), "bands above degree 3 must get zero gradient"
    np.testing.assert_array_equal(np.isfinite(b["dL_dsh"]), True)
    np.testing.assert_array_equal(b["dL_dsh"][:, :16], a["dL_dsh"])
    np.testing.assert_array_equal(a["color"], b["color"])


def test_backward_accepts_the_scratch_of_an_inference_call_and_the_deterministic_mode_says_what_it_needs():
    """Round 5 (GSR_OPT_GRAD_SLABS): an inference call (GSR_FORWARD_INFERENCE) cuts its lists into depth slabs and drops finished
    tiles' pairs -- pairs behind every pixel's last contributor, which the backward never visits -- and gsr_backward walks the slabs'
    segments back to front: the gradients equal those of a full call's scratch up to the order of the atomic sums (a
    well-conditioned scene: 1e-5 of each array's scale).  The deterministic backward sorts ONE list per tile: handed a
    multi-slab scratch it says so instead of reading lists that are laid out differently."""
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import _C
    dev = "cuda:0"
    cloud, cam = scenes.config_c1(P=3000, seed=4), scenes.c1_camera(96, 64)
    c = cloud.to(dev)
    st = settings_for(cam, dev, (0.0, 0.0, 0.0), 1.0, cloud.sh_degree)
    e = torch.Tensor([])
    fwd_args = (st.bg, c.means3D, e, c.opacities, c.scales, c.rotations, 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx,
                st.tanfovy, st.image_height, st.image_width, c.shs, st.sh_degree, st.campos, False, False)
    _C.set_geometry_cache(False)
    _lib.set_option(_lib.OPT_SLABS, 0)
    _lib.set_option(_lib.OPT_SLAB_FIRST, 6)
    _lib.set_option(_lib.OPT_SLAB_MIN_REST, 0)
    grads = {}
    try:
        for inference in (True, False):
            n, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(*fwd_args, inference=inference)
            slabs = len(_C.last_layout()["slab_pairs"])
            assert slabs > 1 if inference else slabs == 1
            bw = lambda: _C.rasterize_gaussians_backward(
                st.bg, c.means3D, radii, e, c.scales, c.rotations, 1.0, e, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy,
                torch.ones_like(color), torch.zeros_like(depth), torch.zeros_like(alpha), c.shs, st.sh_degree, st.campos, geom, n,
                binning, img, alpha, False)
            grads[inference] = [g.clone() for g in bw()]
            torch.cuda.synchronize()
            if inference:
                _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1)
                try:
                    with pytest.raises(RuntimeError, match="ONE list per tile"):
                        bw()
                finally:
                    _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)
        for a, b in zip(grads[True], grads[False]):
            if a.numel():
                assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9
        assert float(grads[True][3].abs().sum()) > 0   # dL_dmeans3D
    finally:
        _C.set_geometry_cache(None)
        _lib.set_option(_lib.OPT_SLABS, 2)
        _lib.set_option(_lib.OPT_SLAB_FIRST, 400)
        _lib.set_option(_lib.OPT_SLAB_MIN_REST, 3000000)


def test_the_deterministic_option_switched_on_between_forward_and_backward_does_not_break_the_graph():
    """every grad-mode forward is a slab call (GSR_OPT_GRAD_SLABS); if GSR_OPT_BACKWARD_DETERMINISTIC is switched on between
    such a forward and its backward the library refuses the scratch.  The autograd node remembers what its forward built
    (``ctx.gsr_slab_forward``): the backward then runs with float atomics, WARNS, and leaves the option as it found it; set before the
    forward the option gives the deterministic backward as always."""
    import warnings
    from autovfx_amd import _lib
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = "cuda:0"
    cloud, cam = scenes.config_c1(P=3000, seed=4), scenes.c1_camera(96, 64)
    c = cloud.to(dev)
    st = settings_for(cam, dev, (0.0, 0.0, 0.0), 1.0, cloud.sh_degree)
    _lib.set_option(_lib.OPT_SLABS, 0)
    _lib.set_option(_lib.OPT_SLAB_FIRST, 6)
    _lib.set_option(_lib.OPT_SLAB_MIN_REST, 0)
    try:
        def run(flip):
            leaves = [t.clone().requires_grad_(True) for t in (c.means3D, c.opacities, c.shs, c.scales, c.rotations)]
            m3, op, sh, sc, rot = leaves
            img, _d, _a, _r = GaussianRasterizer(st)(m3, torch.zeros_like(m3, requires_grad=True), op, shs=sh, scales=sc, rotations=rot)
            if flip:
                _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                img.sum().backward()
            torch.cuda.synchronize()
            return [t.grad.clone() for t in leaves], [str(x.message) for x in w]
        want, quiet = run(False)
        assert not any("DETERMINISTIC" in m for m in quiet)
        got, said = run(True)
        assert any("GSR_OPT_BACKWARD_DETERMINISTIC was switched on after" in m for m in said), said
        assert _lib.get_option(_lib.OPT_BACKWARD_DETERMINISTIC) == 1          # left as the caller set it
        for a, b in zip(got, want):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-9
        det1, said = run(False)                                               # the option is on BEFORE this forward: a full call, no warning
        det2, _ = run(False)
        assert not any("switched on after" in m for m in said)
        for a, b in zip(det1, det2):
            assert torch.equal(a, b)                                          # deterministic: the same bits twice
    finally:
        _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)
        _lib.set_option(_lib.OPT_SLABS, 2)
        _lib.set_option(_lib.OPT_SLAB_FIRST, 400)
        _lib.set_option(_lib.OPT_SLAB_MIN_REST, 3000000)
