"""gsr_backward_raw / the differentiable raw render path (VERDICT round 3, items 3d and "missing" 5): with autograd on,
``render()`` makes ONE rasterizer call from the model's raw tensors and differentiates it with the activations' chain rule
inside the per-Gaussian kernel.  Oracle: the reference's structure run through PyTorch autograd on the same GPU -- exp /
sigmoid / normalize / cat / get_normal in PyTorch, two ``GaussianRasterizer`` calls (``renderer.RAW_AUTOGRAD = False``).
Forward values must be bit-identical.  Gradients:

* where the loss reads ``render`` / ``depth`` only, BOTH paths are held to the truth-based bar of tests/test_backward_gpu.py --
  truth = the fp64 oracle's gradients with respect to the activated parameters, chained through exp / sigmoid / normalize / cat
  by float64 autograd; the reference's fp32 sample = the CPU oracle's fp32 gradients chained by float32 autograd (what the
  reference's own training step computes) -- in both modes of the library;
* where it also reads the normal maps (no CPU oracle for the second pass and the per-pixel post-processing) the two paths are
  compared with each other under GSR_OPT_BACKWARD_DETERMINISTIC, where both are pure functions of the inputs: the comparison
  gives the same numbers on every box.  Bar: max |a - b| <= 2e-4 * max|b| + 1e-6 per gradient tensor.
"""
import numpy as np
import pytest
import torch

from autovfx_amd import _lib, renderer
from autovfx_amd.cameras import orbit_cameras
from oracle import cpu_oracle

from helpers import assert_gradients_vs_truth, report_row
from test_raw_gpu import RENDER_KEYS, raw_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REL, ABS = 2e-4, 1e-6
PARAMS = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")


def leaves(m):
    for k in PARAMS:
        setattr(m, k, getattr(m, k).detach().clone().requires_grad_(True))
    return m


def weights(shape_hw, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    H, W = shape_hw
    r = lambda *s: torch.randn(*s, device=DEV, generator=g) / (H * W)
    return {"render": r(4, H, W), "depth": r(H, W) * 0.1, "normal": r(H, W, 3), "pseudo_normal": r(H, W, 3) * 0.01}


def run(m, cam, bg, wts, raw_autograd, keys, mode="deterministic"):
    saved = renderer.RAW_AUTOGRAD
    renderer.RAW_AUTOGRAD = raw_autograd
    _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1 if mode == "deterministic" else 0)
    try:
        for k in PARAMS:
            getattr(m, k).grad = None
        out = renderer.render(cam, m, renderer.PipelineParams, bg)
        loss = sum((out[k] * wts[k]).sum() for k in keys)
        loss.backward()
        torch.cuda.synchronize()
        grads = {k: getattr(m, k).grad.clone() if getattr(m, k).grad is not None else None for k in PARAMS}
        grads["viewspace_points"] = out["viewspace_points"].grad.clone() if out["viewspace_points"].grad is not None else None
        return {k: out[k].detach().clone() for k in RENDER_KEYS}, grads
    finally:
        renderer.RAW_AUTOGRAD = saved
        _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)


def raw_oracles(m, cam, bg, wts, keys):
    """(reference fp32 sample, fp64 truth) of the gradients with respect to the six RAW tensors for a loss over ``render`` (and
    ``depth``): the CPU oracle differentiates the rasterizer with respect to the activated parameters -- the fp32 tensors the
    model's getters produce on this GPU, the same bits the fused kernels compute -- and autograd carries that through the
    activations (gaussian_model.py:95-128), in float32 for the reference's sample and in float64 for the truth."""
    assert set(keys) <= {"render", "depth"}
    H, W = int(cam.image_height), int(cam.image_width)
    cpu = lambda t: t.detach().cpu()
    with torch.no_grad():
        act = dict(means3D=cpu(m.get_xyz), opacities=cpu(m.get_opacity), scales=cpu(m.get_scaling), rotations=cpu(m.get_rotation),
                   shs=cpu(m.get_features))
    w = cpu(wts["render"])
    kw = dict(act, bg=cpu(bg).numpy(), width=W, height=H, viewmatrix=cpu(cam.world_view_transform), projmatrix=cpu(cam.full_proj_transform),
              campos=cpu(cam.camera_center), tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, sh_degree=m.active_sh_degree,
              dL_dcolor=w[:3].numpy(), dL_dalpha=w[3:4].numpy(),
              dL_ddepth=cpu(wts["depth"])[None].numpy() if "depth" in keys else np.zeros((1, H, W), np.float32))
    out = []
    for grads, dt in ((cpu_oracle.backward(**kw), torch.float32), (cpu_oracle.backward_f64(**kw), torch.float64)):
        raw = {k: cpu(getattr(m, k)).to(dt).requires_grad_(True) for k in PARAMS}
        acts = (raw["_xyz"], torch.exp(raw["_scaling"]), torch.nn.functional.normalize(raw["_rotation"]), torch.sigmoid(raw["_opacity"]),
                torch.cat((raw["_features_dc"], raw["_features_rest"]), dim=1))
        ups = [torch.as_tensor(np.asarray(grads[k])).to(dt).reshape(a.shape) for k, a in
               zip(("dL_dmeans3D", "dL_dscales", "dL_drotations", "dL_dopacity", "dL_dsh"), acts)]
        g = torch.autograd.grad(acts, [raw[k] for k in PARAMS], ups, allow_unused=True)
        d = {k: (torch.zeros_like(raw[k]) if v is None else v).numpy() for k, v in zip(PARAMS, g)}
        d["viewspace_points"] = np.asarray(grads["dL_dmeans2D"])
        out.append(d)
    return out


def compare(name, got, want):
    worst = {}
    for k, b in want.items():
        a = got[k]
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, f"{name}: {k} should carry no gradient"
            continue
        assert a is not None and a.shape == b.shape, f"{name}: {k} has no gradient (or the wrong shape)"
        if b.numel() == 0:
            continue
        scale, err = float(b.abs().max()), float((a - b).abs().max())
        worst[k] = err / max(scale, 1e-30)
        assert err <= REL * scale + ABS, f"{name}: {k} max abs err {err:.3e} vs scale {scale:.3e}"
    report_row("rawgrad:" + name, **worst)


def compare_with_truth(name, m, cam, bg, wts, keys, fused, unfused, mode):
    ref32, truth = raw_oracles(m, cam, bg, wts, keys)
    for tag, got in (("fused", fused), ("structure", unfused)):
        got = {k: v.cpu().numpy() for k, v in got.items() if v is not None}
        assert_gradients_vs_truth(f"raw:{name}:{tag}:{mode}", got, ref32, truth, tuple(got))


@pytest.mark.parametrize("keys", [("render",), ("render", "depth"), ("normal",), ("render", "depth", "normal", "pseudo_normal")])
def test_raw_autograd_matches_pytorch_autograd_through_the_reference_structure(keys):
    """The losses AutoVFX trains with read ``render`` only (scene_representation.py:495-520); the others make sure every
    output the dictionary offers is differentiable with the right values: depth, the normal map (second per-pixel pass +
    chain through get_normal / build_rotation / F.normalize), pseudo normals (through depth)."""
    cam = orbit_cameras(12, 256, 160)[7].to(DEV)
    bg = torch.tensor([0.2, 0.4, 0.1], device=DEV)
    m = leaves(raw_model(25_000, 101, nasty=False))
    wts = weights((160, 256), 3)
    fused_out, fused = run(m, cam, bg, wts, True, keys)
    ref_out, ref = run(m, cam, bg, wts, False, keys)
    for k in RENDER_KEYS:
        assert torch.equal(fused_out[k], ref_out[k]), f"forward {k} differs"
    compare("+".join(keys), fused, ref)
    if set(keys) <= {"render", "depth"}:
        compare_with_truth("+".join(keys), m, cam, bg, wts, keys, fused, ref, "deterministic")
        _, fused_a = run(m, cam, bg, wts, True, keys, mode="atomic")
        _, ref_a = run(m, cam, bg, wts, False, keys, mode="atomic")
        compare_with_truth("+".join(keys), m, cam, bg, wts, keys, fused_a, ref_a, "atomic")
    if "render" in keys:
        assert float(fused["_features_rest"].abs().sum()) > 0 and float(fused["viewspace_points"].abs().sum()) > 0


def test_raw_autograd_with_degenerate_parameters():
    """Tied scales, near-zero and huge quaternions, saturated opacity logits (the `nasty` cases of tests/test_raw_gpu.py):
    gradients through the clamp of F.normalize, sigmoid's flat ends and the argmin's tie rule still match autograd."""
    cam = orbit_cameras(12, 208, 120)[2].to(DEV)
    bg = torch.zeros(3, device=DEV)
    m = leaves(raw_model(12_000, 202, nasty=True))
    wts = weights((120, 208), 9)
    keys = ("render", "normal")
    fused_out, fused = run(m, cam, bg, wts, True, keys)
    ref_out, ref = run(m, cam, bg, wts, False, keys)
    for k in RENDER_KEYS:
        assert torch.equal(fused_out[k], ref_out[k]), f"forward {k} differs"
    compare("nasty", fused, ref)


@pytest.mark.parametrize("degree,M", [(0, 1), (1, 4), (2, 9), (3, 16)])
def test_raw_autograd_sh_layouts(degree, M):
    """_features_rest of every width the reference's models carry ([P,0,3] ... [P,15,3]): the two halves of dL_dsh land in
    the right tensors (LDS-staged store for M = 16, direct stores otherwise)."""
    cam = orbit_cameras(12, 160, 96)[degree].to(DEV)
    bg = torch.tensor([0.5, 0.5, 0.5], device=DEV)
    m = leaves(raw_model(6_000, 300 + degree, sh_degree=degree, M=M, nasty=False))
    wts = weights((96, 160), degree)
    _, fused = run(m, cam, bg, wts, True, ("render",))
    _, ref = run(m, cam, bg, wts, False, ("render",))
    compare(f"sh{degree}", fused, ref)
    assert fused["_features_rest"].shape == (6_000, M - 1, 3) and fused["_features_dc"].shape == (6_000, 1, 3)


def test_training_loop_through_render_lowers_the_loss():
    """A few Adam steps on the raw tensors through render(), as training_3DGS_for_inpainting does: the loss goes down and the
    result equals the same loop run through the reference structure to within optimisation noise."""
    cam = orbit_cameras(12, 192, 120)[4].to(DEV)
    bg = torch.zeros(3, device=DEV)
    target = torch.rand(4, 120, 192, device=DEV)

    def loop(raw_autograd):
        m = leaves(raw_model(8_000, 404, nasty=False))
        opt = torch.optim.Adam([getattr(m, k) for k in PARAMS], lr=5e-3)
        saved = renderer.RAW_AUTOGRAD
        renderer.RAW_AUTOGRAD = raw_autograd
        losses = []
        try:
            for _ in range(10):
                out = renderer.render(cam, m, renderer.PipelineParams, bg)
                loss = (out["render"] - target).abs().mean()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        finally:
            renderer.RAW_AUTOGRAD = saved
        return losses

    a, b = loop(True), loop(False)
    assert a[-1] < a[0] and abs(a[-1] - b[-1]) < 2e-3 * abs(b[0]), (a, b)


def test_c3_full_size_iteration_gradients():
    """BASELINE configs[2] at full size: one training-style iteration through render(), fused raw path against the reference
    structure."""
    from autovfx_amd import scenes
    cam = orbit_cameras(800, 1920, 1080)[400].to(DEV)
    bg = torch.zeros(3, device=DEV)
    m = leaves(raw_model(0, 2, cloud=scenes.config_c3(), nasty=False))
    wts = {"render": torch.randn(4, 1080, 1920, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) / (1080 * 1920)}
    fused_out, fused = run(m, cam, bg, wts, True, ("render",))
    ref_out, ref = run(m, cam, bg, wts, False, ("render",))
    for k in ("render", "depth", "radii"):
        assert torch.equal(fused_out[k], ref_out[k]), k
    compare("c3_full", fused, ref)
    compare_with_truth("c3_full", m, cam, bg, wts, ("render",), fused, ref, "deterministic")
    _, again = run(m, cam, bg, wts, True, ("render",))
    for k, v in fused.items():   # GSR_OPT_BACKWARD_DETERMINISTIC at BASELINE configs[2]: the same bits twice
        assert v is None or torch.equal(v, again[k]), f"{k}: two deterministic runs differ"
    # the default mode: float atomics, and (GSR_OPT_GRAD_SLABS) the forward an inference call in two depth slabs whose segments the
    # backward walks -- same images bit for bit, gradients against the same truth
    slab_out, slab_fused = run(m, cam, bg, wts, True, ("render",), mode="atomic")
    from diff_gaussian_rasterization import _C
    assert len(_C.last_layout()["slab_pairs"]) == 2, "C3 in grad mode should be cut into two depth slabs"
    for k in ("render", "depth", "radii"):
        assert torch.equal(slab_out[k], ref_out[k]), k
    _, slab_ref = run(m, cam, bg, wts, False, ("render",), mode="atomic")
    compare_with_truth("c3_full", m, cam, bg, wts, ("render",), slab_fused, slab_ref, "atomic+slabs")
