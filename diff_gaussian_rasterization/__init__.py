"""Drop-in ``diff_gaussian_rasterization`` for AMD MI355X.

Import-compatible with the package AutoVFX installs from
``sugar/gaussian_splatting/submodules/diff-gaussian-rasterization`` (its Python surface is
``diff_gaussian_rasterization/__init__.py``, cited per item below), so
``sugar/gaussian_splatting/gaussian_renderer/__init__.py::render``,
``scene_representation.py::render_from_3DGS`` and ``sugar/sugar_scene/sugar_model.py`` keep working
unchanged.  Same names, same argument meaning, same 4-tuple ``(color, depth, alpha, radii)``
return, same exceptions for bad argument combinations.  The device side is libgsr_hip.so
(hand-written gfx950 kernels behind the C ABI in include/gsr.h), reached through ``_C``.

Forward and backward are both native (``gsr_forward`` / ``gsr_backward``), so the training-time callers
(``sugar/gaussian_splatting/train.py:84,134``, ``scene_representation.py:495,520`` inpaint re-training) work too.
"""
from __future__ import annotations

import functools
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _C

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]

_SNAPSHOT_FW = "snapshot_fw.dump"
_SNAPSHOT_BW = "snapshot_bw.dump"


class GaussianRasterizationSettings(NamedTuple):
    """Per-camera configuration (reference ``__init__.py:160-172``); field order is part of the API."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _host_copy(values):
    return tuple(v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v for v in values)


def _call_with_snapshot(fn, args, debug: bool, dump_path: str, what: str):
    """In debug mode keep a host copy of the inputs and dump it if the device call raises
    (reference ``__init__.py:83-90,135-142``)."""
    if not debug:
        return fn(*args)
    saved = _host_copy(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_path)
        print(f"\nAn error occured in {what}. Please forward {dump_path} for debugging.")
        raise


class _RasterizeGaussians(torch.autograd.Function):
    """Autograd node (reference ``__init__.py:44-158``)."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        s = raster_settings
        # positional order of DGR/rasterize_points.h:18-38
        native_args = (s.bg, means3D, colors_precomp, opacities, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                       s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, sh,
                       s.sh_degree, s.campos, s.prefiltered, s.debug)
        # Nothing upstream wants a gradient (torch.no_grad(), or plain tensors): the binding may say so, and the library
        # then builds its lists the cheaper way (include/gsr.h: GSR_FORWARD_INFERENCE); same images, same radii.
        # ... and a call that WILL be differentiated may be built that way too (GSR_OPT_GRAD_SLABS): the backward walks the depth
        # slabs' list segments; only the deterministic backward needs one list per tile, i.e. a full call.
        inference = not any(ctx.needs_input_grad) or _C.grad_slabs()
        forward_fn = functools.partial(_C.rasterize_gaussians, inference=inference)
        (num_rendered, color, depth, alpha, radii, geom_buffer, binning_buffer, img_buffer) = _call_with_snapshot(
            forward_fn, native_args, s.debug, _SNAPSHOT_FW, "forward")
        ctx.raster_settings = s
        ctx.num_rendered = num_rendered
        ctx.gsr_slab_forward = bool(inference)      # the lists of this call may be per depth slab: see backward
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buffer,
                              binning_buffer, img_buffer, alpha)
        ctx.mark_non_differentiable(radii)
        # an output the loss never touched arrives in backward as None instead of a zero-filled image: the native backward
        # then skips the depth / alpha terms altogether (include/gsr.h: gsr_backward)
        ctx.set_materialize_grads(False)
        return color, depth, alpha, radii

    @staticmethod
    def backward(ctx, grad_out_color, grad_out_depth, grad_out_alpha, _grad_radii):
        s = ctx.raster_settings
        (colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom_buffer, binning_buffer,
         img_buffer, alpha) = ctx.saved_tensors
        if grad_out_color is None and grad_out_depth is None and grad_out_alpha is None:
            return (None,) * 9
        if grad_out_color is None:
            grad_out_color = torch.zeros((3, alpha.shape[1], alpha.shape[2]), dtype=alpha.dtype, device=alpha.device)
        # positional order of DGR/rasterize_points.h:40-65
        native_args = (s.bg, means3D, radii, colors_precomp, scales, rotations, s.scale_modifier, cov3Ds_precomp,
                       s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, grad_out_color, grad_out_depth,
                       grad_out_alpha, sh, s.sh_degree, s.campos, geom_buffer, ctx.num_rendered, binning_buffer,
                       img_buffer, alpha, s.debug)
        with _C.deterministic_backward_guard(ctx.gsr_slab_forward):
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
             grad_rotations) = _call_with_snapshot(functools.partial(_C.rasterize_gaussians_backward, skip_unused=True), native_args,
                                                   s.debug, _SNAPSHOT_BW, "backward")
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    """Functional entry point (reference ``__init__.py:21-42``)."""
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


def _absent() -> torch.Tensor:
    # The native layer recognises "not provided" as an empty tensor (reference __init__.py:200-210).
    return torch.Tensor([])


class GaussianRasterizer(nn.Module):
    """Module wrapper (reference ``__init__.py:174-223``)."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of points in front of the near plane (view-space z > 0.2)."""
        s = self.raster_settings
        with torch.no_grad():
            return _C.mark_visible(positions, s.viewmatrix, s.projmatrix)

    def forward(self, means3D, means2D, opacities, shs: Optional[torch.Tensor] = None,
                colors_precomp: Optional[torch.Tensor] = None, scales: Optional[torch.Tensor] = None,
                rotations: Optional[torch.Tensor] = None, cov3D_precomp: Optional[torch.Tensor] = None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        has_any_sr = scales is not None or rotations is not None
        has_full_sr = scales is not None and rotations is not None
        if (not has_full_sr and cov3D_precomp is None) or (has_any_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        fill = lambda t: _absent() if t is None else t
        return rasterize_gaussians(means3D, means2D, fill(shs), fill(colors_precomp), opacities, fill(scales),
                                   fill(rotations), fill(cov3D_precomp), self.raster_settings)
