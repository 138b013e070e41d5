"""``render()``: the per-frame function AutoVFX calls, mirrored for the MI355X rasterizer.

Same signature, same result dictionary and the same arithmetic as
``sugar/gaussian_splatting/gaussian_renderer/__init__.py:83-218``:

1. activations and SH / normal preparation in PyTorch (``:118-146,169-171``),
2. rasterizer pass 1, SH colours (``:151-159``) -> RGBA (``:162``) and depth,
3. rasterizer pass 2, per-Gaussian normals as precomputed colours (``:176-184``) -> normal map,
   normalised per pixel (``:189-194``),
4. pseudo-normals from the depth map by local differences (``:197-208`` with ``depth_pcd2normal`` ``:22-38`` and
   ``get_ray_directions`` ``:41-80``; ``kornia.create_meshgrid`` is replaced by the two ``arange``s it amounts to).

It exists for two reasons: the reference's own ``render`` cannot be imported where ``kornia``,
``plyfile`` and ``simple_knn`` are missing, and it is the boundary at which the second of the two
measurements of SURVEY.md section 8d is taken (``bench.py --boundary render``).  The second rasterizer pass
hits the geometry cache of ``diff_gaussian_rasterization._C`` (same tensors, new colours), so it costs
one blend launch instead of a full pipeline; results are bit-identical either way.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from .cameras import fov2focal


class PipelineParams:
    """``arguments.PipelineParams`` defaults (``sugar/sugar_scene/gs_model.py:34-37``)."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


# False: render() keeps the reference's structure even without autograd -- ~40 PyTorch launches around two rasterizer calls
# (gaussian_renderer/__init__.py:118-208) -- instead of the two fused kernels and the folded normal pass.  bench.py times
# that shape next to the default; results are the same images.
FUSE_ELEMENTWISE = True

# True: a model that exposes the reference's six raw parameter tensors with the reference's activations (any
# ``scene.gaussian_model.GaussianModel``: ``_xyz, _scaling, _rotation, _opacity, _features_dc, _features_rest``) is rendered
# from those tensors directly (``gsr_forward_raw``): exp / sigmoid / normalize, the SH concat and the per-Gaussian view normal
# happen inside the projection and colour kernels instead of ~25 PyTorch launches and 2.3 GB of traffic per frame at 3 M
# Gaussians.  No memo, nothing cached between frames: parameters may change every frame (training, dynamic scenes).
# Bit-identical to the getters path on this GPU (tests/test_raw_gpu.py).
RAW_PARAMETERS = True
_RAW_ATTRS = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")
_RAW_ACTIVATIONS = (("scaling_activation", torch.exp), ("opacity_activation", torch.sigmoid),
                    ("rotation_activation", torch.nn.functional.normalize))
_GETTERS = ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features", "get_normal")
_getters_verdict: dict = {}   # type -> bool


def _getters_are_one_class_s(cls) -> bool:
    """A subclass or wrapper that overrides ONE of the getters ``render()`` would otherwise call (a masked opacity, a transformed
    xyz, its own get_normal) while keeping the raw attributes must not be rendered from the raw tensors: that would bypass the
    override without a word.  The raw path is taken only when all six getters come from ONE class of the model's MRO -- the class
    that defines the model (the reference's GaussianModel, this package's, a test double) -- i.e. none of them was overridden
    further down.  A model that overrides them all and still wants the raw path says so with ``gsr_raw_parameters = True`` on its
    class; one that wants out says ``False``.  (Cached per type.)"""
    v = _getters_verdict.get(cls)
    if v is None:
        owners = set()
        for name in _GETTERS:
            owner = next((c for c in cls.__mro__ if name in c.__dict__), None)
            if owner is None:
                continue          # (a model without get_normal renders no normals anyway: the getter is simply never called)
            owners.add(owner)
        v = len(owners) <= 1
        _getters_verdict[cls] = v
    return v


# True: with autograd ON such a model is rendered by ONE full rasterizer call from the raw tensors as well (colour + normal
# image in the same walk of the lists) and differentiated by gsr_backward_raw, which applies the activations' chain rule inside
# the per-Gaussian kernel: a training iteration (scene_representation.py:495-520, train.py:84-134) runs one forward and one
# backward pass of the rasterizer instead of ~25 PyTorch activation kernels, two forward passes and their autograd graph.
# Same values forward (bit-identical images), gradients within the tolerance of sums formed with atomics
# (tests/test_raw_autograd_gpu.py).  False: the reference's structure (PyTorch activations, two rasterizer calls).
RAW_AUTOGRAD = True
# True: in grad mode the normal and pseudo-normal maps take their values from the fused kernel and build their autograd graph only
# when a gradient actually arrives for them (_NormalMaps).  False: the PyTorch expressions run in every forward.
LAZY_NORMAL_GRADIENTS = True


class _RasterizeRaw(torch.autograd.Function):
    """``gsr_forward_raw`` (a full call) / ``gsr_backward_raw`` as one autograd node over the model's six raw tensors."""

    @staticmethod
    def forward(ctx, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, means2D, settings):
        from diff_gaussian_rasterization import _C
        s = settings
        inference = not any(ctx.needs_input_grad) or _C.grad_slabs()   # (GSR_OPT_GRAD_SLABS: the backward walks the slabs' segments)
        (n, color, depth, alpha, radii, geom, binning, image, normal) = _C.rasterize_gaussians_raw(
            s.bg, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, s.scale_modifier, s.viewmatrix, s.projmatrix,
            s.tanfovx, s.tanfovy, s.image_height, s.image_width, s.sh_degree, s.campos, s.prefiltered, s.debug, want_normal=True,
            inference=inference)
        ctx.settings, ctx.num_rendered = s, n
        ctx.gsr_slab_forward = bool(inference)     # (diff_gaussian_rasterization._C.deterministic_backward_guard)
        # colour and alpha are planes of ONE output buffer: the node hands out that buffer as the [4,H,W] RGBA image render() returns
        # (the reference's torch.cat((rendered, alpha)), :186, without the copy and without the concat's graph node)
        rgba = _C.rgba_planes(color, alpha)
        ctx.save_for_backward(xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, radii, geom, binning, image, rgba)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)   # an image the loss never read arrives as None: its pass / terms are skipped
        return rgba, depth, radii, normal

    @staticmethod
    def backward(ctx, g_rgba, g_depth, _g_radii, g_normal):
        from diff_gaussian_rasterization import _C
        if g_rgba is None and g_depth is None and g_normal is None:
            return (None,) * 8
        s = ctx.settings
        xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, radii, geom, binning, image, rgba = ctx.saved_tensors
        alpha = rgba[3:4]
        g_color = g_alpha = None
        if g_rgba is not None:
            g_rgba = g_rgba.contiguous()
            g_color, g_alpha = g_rgba[:3], g_rgba[3:4]
        with _C.deterministic_backward_guard(ctx.gsr_slab_forward):
            g2d, gxyz, gls, grot, gop, gdc, grest = _C.rasterize_gaussians_raw_backward(
                s.bg, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, radii, s.scale_modifier, s.viewmatrix,
                s.projmatrix, s.tanfovx, s.tanfovy, g_color, g_depth, g_alpha, g_normal, s.sh_degree, s.campos, geom, ctx.num_rendered,
                binning, image, alpha, s.debug)
        return gxyz, gls, grot, gop, gdc, grest, g2d, None


def _normal_maps_torch(normal_image, depth_image, c2w, h, w, fx, fy):
    """The reference's per-pixel post-processing (gaussian_renderer/__init__.py:186-208) as PyTorch expressions."""
    normal = (normal_image - 0.5) * 2.0
    normal = torch.nn.functional.normalize(normal.permute(1, 2, 0), p=2, dim=-1)
    directions = get_ray_directions(h, w, fx, fy, w / 2, h / 2, depth_image.device)
    rays_d = directions @ c2w[:3, :3].T
    rays_o = c2w[:3, 3].expand_as(rays_d)
    return normal, depth_pcd2normal(rays_o + rays_d * depth_image.unsqueeze(-1))


class _NormalMaps(torch.autograd.Function):
    """The normal map and the pseudo-normal map of a grad-mode ``render()``: VALUES from the fused kernel the inference path uses
    (``gsr_normal_maps``: the same formulas in the same order, ``torch.equal`` to the PyTorch expressions -- tests/test_raw_gpu.py),
    GRADIENTS, if the loss ever asks for them, by re-running the PyTorch expressions under autograd in ``backward``.  The
    reference's training losses read ``render`` only (scene_representation.py:507-510, train.py:84-134): they used to pay ~15
    PyTorch launches and their graph per iteration for two maps nobody differentiated."""

    @staticmethod
    def forward(ctx, normal_image, depth_image, c2w, h, w, fx, fy):
        ctx.save_for_backward(normal_image, depth_image, c2w)
        ctx.dims = (h, w, fx, fy)
        ctx.set_materialize_grads(False)   # a map the loss never read arrives as None
        normal, pseudo = _fused_normal_maps(normal_image.detach(), depth_image.detach(), c2w, fx, fy, w / 2, h / 2)
        return normal, pseudo

    @staticmethod
    def backward(ctx, g_normal, g_pseudo):
        normal_image, depth_image, c2w = ctx.saved_tensors
        h, w, fx, fy = ctx.dims
        with torch.enable_grad():
            n = normal_image.detach().requires_grad_(True)
            d = depth_image.detach().requires_grad_(True)
            normal, pseudo = _normal_maps_torch(n, d, c2w, h, w, fx, fy)
            outs, grads = [], []
            if g_normal is not None:
                outs.append(normal); grads.append(g_normal)
            if g_pseudo is not None:
                outs.append(pseudo); grads.append(g_pseudo)
            gn, gd = torch.autograd.grad(outs, [n, d], grads, allow_unused=True) if outs else (None, None)
        return gn, gd, None, None, None, None, None


def raw_parameters(pc):
    """The six raw tensors of ``pc`` when rendering from them is the same as rendering through its getters, else None:
    all six attributes are float32 tensors on one GPU with consistent shapes, and the model's activation functions -- the
    reference keeps them as attributes (``setup_functions``, ``scene/gaussian_model.py:25-45``) -- are the stock ones, and no
    subclass overrode one of the getters the raw path would bypass (``_getters_are_one_class_s``).  A model can opt out with
    ``pc.gsr_raw_parameters = False``."""
    if not getattr(pc, "gsr_raw_parameters", True):
        return None
    if "gsr_raw_parameters" not in type(pc).__dict__ and not _getters_are_one_class_s(type(pc)):
        return None
    try:
        t = tuple(getattr(pc, a) for a in _RAW_ATTRS)
    except AttributeError:
        return None
    if not all(isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.device == t[0].device for x in t):
        return None
    for attr, fn in _RAW_ACTIVATIONS:
        if getattr(pc, attr, fn) is not fn:
            return None
    xyz, ls, rot, op, dc, rest = t
    P = xyz.shape[0]
    if (xyz.dim() != 2 or tuple(xyz.shape) != (P, 3) or tuple(ls.shape) != (P, 3) or tuple(rot.shape) != (P, 4) or op.numel() != P
            or tuple(dc.shape) != (P, 1, 3) or rest.dim() != 3 or rest.shape[0] != P or rest.shape[2] != 3):
        return None
    return t


def _stream_ptr(device):
    import ctypes
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _fused_view_normals(xyz: torch.Tensor, axis: torch.Tensor, campos: torch.Tensor) -> torch.Tensor:
    """``pc.get_normal(normalize(xyz - campos)) * 0.5 + 0.5`` in one kernel."""
    from . import _lib
    xyz_, axis_, cp_ = xyz.contiguous(), axis.contiguous().float(), campos.contiguous().float()
    out = torch.empty_like(xyz_)
    with torch.cuda.device(xyz.device):
        rc = _lib.lib.gsr_view_normals(int(xyz_.shape[0]), xyz_.data_ptr(), axis_.data_ptr(), cp_.data_ptr(), out.data_ptr(),
                                       _stream_ptr(xyz.device))
    if rc != 0:
        raise RuntimeError(f"gsr_view_normals failed ({rc}): {_lib.last_error()}")
    return out


def _fused_normal_maps(normal_rgb: torch.Tensor, depth: torch.Tensor, c2w: torch.Tensor, fx, fy, cx, cy):
    """Normal map [H,W,3] from the raw pass-2 image and pseudo-normal map [H,W,3] from the depth map, one kernel."""
    from . import _lib
    n_, d_, m_ = normal_rgb.contiguous(), depth.contiguous(), c2w.contiguous().float()
    H, W = int(d_.shape[0]), int(d_.shape[1])
    normal = torch.empty((H, W, 3), dtype=torch.float32, device=d_.device)
    pseudo = torch.empty((H, W, 3), dtype=torch.float32, device=d_.device)
    with torch.cuda.device(d_.device):
        rc = _lib.lib.gsr_normal_maps(W, H, n_.data_ptr(), d_.data_ptr(), m_.data_ptr(), float(fx), float(fy), float(cx),
                                      float(cy), normal.data_ptr(), pseudo.data_ptr(), _stream_ptr(d_.device))
    if rc != 0:
        raise RuntimeError(f"gsr_normal_maps failed ({rc}): {_lib.last_error()}")
    return normal, pseudo


def depth_pcd2normal(xyz: torch.Tensor) -> torch.Tensor:
    """Un-projected points [H,W,3] -> pseudo normal map by central differences (``:22-38``)."""
    hd, wd, _ = xyz.shape
    bottom, top = xyz[2:hd, 1:wd - 1, :], xyz[0:hd - 2, 1:wd - 1, :]
    right, left = xyz[1:hd - 1, 2:wd, :], xyz[1:hd - 1, 0:wd - 2, :]
    n = torch.cross(right - left, top - bottom, dim=-1)
    n = torch.nn.functional.normalize(n, p=2, dim=-1)
    return torch.nn.functional.pad(n.permute(2, 0, 1), (1, 1, 1, 1), mode="constant").permute(1, 2, 0)


def get_ray_directions(H: int, W: int, fx: float, fy: float, cx: float, cy: float, device) -> torch.Tensor:
    """Camera-space ray through each pixel centre, [H,W,3] (``:41-80`` with ``random=False``).  The intrinsics enter the
    arithmetic as the 0-dim float32 CPU tensors the reference slices out of ``torch.FloatTensor([[fx, 0, cx], ...])``
    (``:199``): on the GPU a division by such a scalar is a multiplication by ``float(1.0 / double(float32(fx)))``, which
    a Python float in their place would not reproduce to the last bit."""
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=device),
                          torch.arange(W, dtype=torch.float32, device=device), indexing="ij")
    fx_, fy_, cx_, cy_ = (torch.tensor(float(t), dtype=torch.float32) for t in (fx, fy, cx, cy))
    return torch.stack(((u - cx_ + 0.5) / fx_, (v - cy_ + 0.5) / fy_, torch.ones_like(u)), -1)


def sh_basis(deg: int, dirs: torch.Tensor) -> torch.Tensor:
    """Real spherical-harmonics basis of bands 0..deg at unit directions ``dirs[N,3]`` -> ``[N,(deg+1)^2]``, with the
    sign / ordering convention of the rasterizer's constants (``DGR/cuda_rasterizer/auxiliary.h:22-39``,
    ``utils/sh_utils.py:24-55``).  Bands above 3 are not evaluated by the rasterizer and not offered here."""
    if not 0 <= deg <= 3:
        raise ValueError("sh_basis: degree must be 0..3")
    x, y, z = dirs.unbind(-1)
    cols = [torch.full_like(x, 0.28209479177387814)]
    if deg > 0:
        c1 = 0.4886025119029199
        cols += [-c1 * y, c1 * z, -c1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        cols += [1.0925484305920792 * xy, -1.0925484305920792 * yz, 0.31539156525252005 * (2.0 * zz - xx - yy),
                 -1.0925484305920792 * xz, 0.5462742152960396 * (xx - yy)]
    if deg > 2:
        cols += [-0.5900435899266435 * y * (3 * xx - yy), 2.890611442640554 * xy * z,
                 -0.4570457994644658 * y * (4 * zz - xx - yy), 0.3731763325901154 * z * (2 * zz - 3 * xx - 3 * yy),
                 -0.4570457994644658 * x * (4 * zz - xx - yy), 1.445305721320277 * z * (xx - yy),
                 -0.5900435899266435 * x * (xx - 3 * yy)]
    return torch.stack(cols, dim=-1)


def sh_to_rgb_python(deg: int, features: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """What ``render()`` does with ``pipe.convert_SHs_python`` (``gaussian_renderer/__init__.py:141-146``):
    ``clamp_min(eval_sh(deg, shs, dir) + 0.5, 0)`` with ``features[N,M,3]`` in the rasterizer's layout, evaluated as one
    basis matrix times the coefficients (differentiable; sums in a different order than ``eval_sh``: last-ulp
    differences).  ``deg`` above 3 evaluates bands 0..3, like the rasterizer."""
    deg = min(int(deg), 3)
    k = (deg + 1) ** 2
    basis = sh_basis(deg, dirs)                                   # [N,k]
    return torch.clamp_min((features[:, :k, :] * basis.unsqueeze(-1)).sum(dim=1) + 0.5, 0.0)


class PendingRender:
    """A ``render`` call whose rasterizer call is half queued (``render_begin``); ``finish()`` queues the rest and
    returns the dict ``render`` returns."""

    def __init__(self, tail, ready=None):
        self._tail, self._ready = tail, ready

    def ready(self) -> bool:
        """True when ``finish()`` will not wait for the GPU."""
        return True if self._ready is None else self._ready()

    def finish(self):
        if self._tail is None:
            raise RuntimeError("PendingRender.finish() called twice")
        tail, self._tail = self._tail, None
        return tail()


def render_begin(viewpoint_camera, pc, pipe=PipelineParams, bg_color: Optional[torch.Tensor] = None,
                 scaling_modifier: float = 1.0, override_color: Optional[torch.Tensor] = None) -> PendingRender:
    """``render`` in two halves for inference loops that keep several frames in flight from one host thread
    (``frame_parallel.render_shard(driver="pipelined")``): the per-Gaussian normals, the projection and the depth
    sort are queued on the current stream and the call returns without waiting for the GPU; ``finish()`` (same
    thread, same current stream) queues the rest.  Needs what the fused single-pass path needs: autograd off, data on
    the GPU, a model with the six raw parameter tensors or with ``get_minimum_axis``.  Images are those of ``render``, bit for bit."""
    out = _render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, split=True)
    if not isinstance(out, PendingRender):
        raise RuntimeError("render_begin needs torch.no_grad(), float32 data on the GPU and a model with the raw parameter "
                           "tensors or get_minimum_axis")
    return out


def render(viewpoint_camera, pc, pipe=PipelineParams, bg_color: Optional[torch.Tensor] = None,
           scaling_modifier: float = 1.0, override_color: Optional[torch.Tensor] = None):
    """Render one view.  ``pc`` is anything with the ``GaussianModel`` getters; ``bg_color`` lives on the GPU."""
    return _render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, split=False)


def _render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color, split):
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

    xyz = pc.get_xyz
    if torch.is_grad_enabled():
        # (the reference writes ``torch.zeros_like(..., requires_grad=True) + 0`` and retain_grad(): a zero fill, an add and a graph
        # node per frame for a tensor that is only ever read through ``.grad``; a zero leaf carries the same values and the same .grad)
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True)
    else:   # only ever read through its gradient (densification statistics): without autograd it is P x 3 zeros that
        # nobody reads, returned as a broadcast view instead of a fresh 12-byte-per-Gaussian fill per frame
        screenspace_points = torch.zeros((1, 1), dtype=xyz.dtype, device=xyz.device).expand(xyz.shape)
    if screenspace_points.requires_grad:   # (without autograd retain_grad() only raises: 0.16 ms per frame of exception handling)
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
    settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    # (the module object is only needed on the path that makes two separate rasterizer calls: constructing an nn.Module per frame
    # costs 20 us of host time that the fused paths, which call the binding directly, need not pay)

    h, w = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
    fx, fy = fov2focal(viewpoint_camera.FoVx, w), fov2focal(viewpoint_camera.FoVy, h)
    c2w = getattr(viewpoint_camera, "view_world_transform", None)   # formed with the camera when it is ours
    if c2w is None:
        c2w = viewpoint_camera.world_view_transform.inverse()

    # Straight from the model's raw parameters (gsr_forward_raw): no getter is called, nothing is activated in PyTorch.
    raw = None
    if (RAW_PARAMETERS and FUSE_ELEMENTWISE and override_color is None and not pipe.convert_SHs_python
            and not pipe.compute_cov3D_python and (RAW_AUTOGRAD or not torch.is_grad_enabled())):
        raw = raw_parameters(pc)
    if raw is not None and torch.is_grad_enabled():
        # one differentiable rasterizer call from the raw tensors; the per-pixel post-processing stays in PyTorch (:186-208),
        # so whatever the loss reads -- RGBA, depth, normal map, pseudo normals -- carries its gradient back
        rendered_image, depth_image, radii, normal_image = _RasterizeRaw.apply(*raw, screenspace_points, settings)
        depth_image = depth_image.squeeze(0)
        if LAZY_NORMAL_GRADIENTS:
            normal_image, pseudo_normal = _NormalMaps.apply(normal_image, depth_image, c2w, h, w, fx, fy)
        else:
            normal_image, pseudo_normal = _normal_maps_torch(normal_image, depth_image, c2w, h, w, fx, fy)
        return {"render": rendered_image, "depth": depth_image, "normal": normal_image, "pseudo_normal": pseudo_normal,
                "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
    if raw is not None:
        from diff_gaussian_rasterization import _C
        s_ = settings
        raw_args = (s_.bg, *raw, s_.scale_modifier, s_.viewmatrix, s_.projmatrix, s_.tanfovx, s_.tanfovy, s_.image_height,
                    s_.image_width, s_.sh_degree, s_.campos, s_.prefiltered, s_.debug)

        def assemble_raw(result):
            (_n, rendered_image, depth_image, alpha_image, radii, _g, _b, _i, normal_image) = result
            rendered_image = _C.rgba_planes(rendered_image, alpha_image)
            depth_image = depth_image.squeeze(0)
            normal_image, pseudo_normal = _fused_normal_maps(normal_image, depth_image, c2w, fx, fy, w / 2, h / 2)
            return {"render": rendered_image, "depth": depth_image, "normal": normal_image,
                    "pseudo_normal": pseudo_normal, "viewspace_points": screenspace_points,
                    "visibility_filter": radii > 0, "radii": radii}

        if split:
            pending = _C.rasterize_gaussians_raw_begin(*raw_args, want_normal=True, inference=True)
            return PendingRender(lambda: assemble_raw(pending.finish()), pending.ready)
        return assemble_raw(_C.rasterize_gaussians_raw(*raw_args, want_normal=True, inference=True))

    means3D, means2D, opacity = xyz, screenspace_points, pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    # With autograd off and the data on the GPU, the elementwise work around the passes runs as two fused kernels
    # (gsr_view_normals / gsr_normal_maps) instead of ~40 PyTorch launches, and the normal pass is folded into the
    # first one (gsr_forward_extra); same formulas, same images (tests).
    fused = FUSE_ELEMENTWISE and (not torch.is_grad_enabled()) and xyz.is_cuda and xyz.dtype == torch.float32 and hasattr(pc, "get_minimum_axis")
    dir_pp_normalized = None
    if not fused or (override_color is None and pipe.convert_SHs_python):
        dir_pp = xyz - viewpoint_camera.camera_center.repeat(xyz.shape[0], 1)
        dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:   # :141-146: colours from the SH in PyTorch, handed over as colors_precomp
            colors_precomp = sh_to_rgb_python(pc.active_sh_degree, pc.get_features, dir_pp_normalized)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    if fused:
        # One pass: the per-Gaussian normals ride through the SAME walk of the per-tile lists as a second feature
        # set (gsr_forward_extra); the normal image is what the reference's second pass (:176-184) returns, bit for bit.
        from diff_gaussian_rasterization import _C
        normal_normed = _fused_view_normals(xyz, pc.get_minimum_axis, viewpoint_camera.camera_center)
        absent = torch.Tensor([])
        s_ = settings
        call_args = (
            s_.bg, means3D, absent if colors_precomp is None else colors_precomp, opacity,
            absent if scales is None else scales, absent if rotations is None else rotations, s_.scale_modifier,
            absent if cov3D_precomp is None else cov3D_precomp, s_.viewmatrix, s_.projmatrix, s_.tanfovx, s_.tanfovy,
            s_.image_height, s_.image_width, absent if shs is None else shs, s_.sh_degree, s_.campos, s_.prefiltered,
            s_.debug, normal_normed)

        def assemble(result):
            (_n, rendered_image, depth_image, alpha_image, radii, _g, _b, _i, normal_image) = result
            rendered_image = _C.rgba_planes(rendered_image, alpha_image)   # torch.cat((colour, alpha)) without the copy
            depth_image = depth_image.squeeze(0)
            normal_image, pseudo_normal = _fused_normal_maps(normal_image, depth_image, c2w, fx, fy, w / 2, h / 2)
            return {"render": rendered_image, "depth": depth_image, "normal": normal_image,
                    "pseudo_normal": pseudo_normal, "viewspace_points": screenspace_points,
                    "visibility_filter": radii > 0, "radii": radii}

        if split:
            pending = _C.rasterize_gaussians_begin(*call_args, inference=True)   # (fused: autograd is off)
            return PendingRender(lambda: assemble(pending.finish()), pending.ready)
        return assemble(_C.rasterize_gaussians_extra(*call_args, inference=True))
    else:
        rasterizer = GaussianRasterizer(raster_settings=settings)
        rendered_image, depth_image, alpha_image, radii = rasterizer(
            means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity, scales=scales,
            rotations=rotations, cov3D_precomp=cov3D_precomp)
        rendered_image = torch.cat((rendered_image, alpha_image), dim=0)
        depth_image = depth_image.squeeze(0)
        normal_normed = pc.get_normal(dir_pp_normalized=dir_pp_normalized) * 0.5 + 0.5
        normal_image = rasterizer(means3D=means3D, means2D=means2D, shs=None, colors_precomp=normal_normed,
                                  opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)[0]
        normal_image = (normal_image - 0.5) * 2.0
        normal_image = torch.nn.functional.normalize(normal_image.permute(1, 2, 0), p=2, dim=-1)
        directions = get_ray_directions(h, w, fx, fy, w / 2, h / 2, depth_image.device)
        rays_d = directions @ c2w[:3, :3].T
        rays_o = c2w[:3, 3].expand_as(rays_d)
        points3D = rays_o + rays_d * depth_image.unsqueeze(-1)
        pseudo_normal = depth_pcd2normal(points3D)

    return {"render": rendered_image, "depth": depth_image, "normal": normal_image, "pseudo_normal": pseudo_normal,
            "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
