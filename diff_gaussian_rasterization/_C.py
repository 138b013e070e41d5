"""``diff_gaussian_rasterization._C`` -- the extension-module surface of the reference, backed by
libgsr_hip.so through its C ABI (include/gsr.h).

The reference builds ``_C`` as a pybind11/CUDA torch extension exposing three functions
(``DGR/ext.cpp:16-18``, DGR = sugar/gaussian_splatting/submodules/diff-gaussian-rasterization):

* ``rasterize_gaussians``          -> ``RasterizeGaussiansCUDA``         ``DGR/rasterize_points.cu:36-119``
* ``rasterize_gaussians_backward`` -> ``RasterizeGaussiansBackwardCUDA`` ``DGR/rasterize_points.cu:121-209``
* ``mark_visible``                 -> ``markVisible``                    ``DGR/rasterize_points.cu:211-230``

This module keeps those names, argument orders, return tuples and error behaviour; what changes
is everything underneath (HIP kernels for gfx950, called with raw device pointers on torch's
current HIP stream).  Tensors must live on a GPU: there is no CPU path, by design.
"""
from __future__ import annotations

import ctypes
import threading
from typing import Optional, Tuple

import torch

from autovfx_amd import _lib

_tls = threading.local()


def _alloc_geom(nbytes, _user):
    return _tls.call.alloc("geom", nbytes)


def _alloc_binning(nbytes, _user):
    return _tls.call.alloc("binning", nbytes)


def _alloc_image(nbytes, _user):
    return _tls.call.alloc("image", nbytes)


# ctypes trampolines are created once; the per-call state lives in thread-local storage.
_GEOM_CB = _lib.ALLOC_FN(_alloc_geom)
_BINNING_CB = _lib.ALLOC_FN(_alloc_binning)
_IMAGE_CB = _lib.ALLOC_FN(_alloc_image)


# ---- test hook: what freshly allocated scratch / output / gradient tensors contain --------------------------------
# Every tensor this module hands to the library is ``torch.empty``: the library must write what it later reads, whatever
# the allocator recycled.  ``set_alloc_poison`` makes that property testable (tests/test_poison_gpu.py): None = leave the
# memory as the allocator returned it (the default), an int 0..255 = fill with that byte (0xFF: every float a NaN, every
# index 4 G), "random" = random bytes.  Results must not depend on it.
_POISON = None


def set_alloc_poison(pattern) -> None:
    global _POISON
    if pattern is not None and pattern != "random" and not (isinstance(pattern, int) and 0 <= pattern <= 255):
        raise ValueError("poison pattern: None, 'random' or a byte value")
    _POISON = pattern


def _new(shape, dtype, device) -> torch.Tensor:
    t = torch.empty(shape, dtype=dtype, device=device)
    if _POISON is not None and t.numel():
        raw = t.view(torch.uint8) if t.dim() else t.reshape(1).view(torch.uint8)
        if _POISON == "random":
            raw.random_(0, 256)
        else:
            raw.fill_(_POISON)
    return t


class _CallScratch:
    """The three growable byte tensors of ``rasterize_points.cu:73-80`` (resizeFunctional)."""

    def __init__(self, device: torch.device):
        self.device = device
        self.buffers = {k: torch.empty(0, dtype=torch.uint8, device=device) for k in ("geom", "binning", "image")}

    def alloc(self, which: str, nbytes: int) -> int:
        t = _new(int(nbytes), torch.uint8, self.device)
        self.buffers[which] = t
        return t.data_ptr()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Device pointer of a tensor, or NULL for the reference's "empty tensor means absent"."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def _f32c(name: str, t: torch.Tensor, device: torch.device) -> torch.Tensor:
    if t.numel() == 0:
        return t
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected a float32 tensor, got {t.dtype}")
    if t.device != device:
        raise RuntimeError(f"{name}: expected a tensor on {device}, got {t.device}")
    return t.contiguous()


def _require_gpu(t: torch.Tensor, what: str) -> torch.device:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} is on {t.device}: the MI355X rasterizer only runs on a HIP device (there is no CPU fallback)")
    return t.device


# ---- geometry reuse between the two passes of render() (OPT-IN) ------------------------------------------------------
# gaussian_renderer.render() calls the rasterizer twice per frame with the very same tensor objects for the
# geometry (means3D, opacity, scales, rotations / cov3D, camera) and only the colour source changed
# (gaussian_renderer/__init__.py:151-159 then :176-184).  With the reuse switched on, a call whose geometry inputs are the
# SAME tensor objects at the SAME autograd version as the call just before it on this thread and stream, and which brings
# precomputed colours, reuses everything up to the per-tile lists: only the blend kernel runs (bit-identical images, 12 % of
# an unchanged render() at C3).
#
# It is OFF unless asked for -- ``GSR_GEOMETRY_CACHE=1`` in the environment or ``set_geometry_cache(True)`` -- because it
# is a behaviour the reference does not have: a write that bypasses the version counter between the two calls
# (`t.data.add_()`, a raw-pointer write from another extension, a DLPack alias) is invisible to it, and the second call
# would composite over the first call's geometry.  The reference recomputes; by default so does this module.
# (autovfx_amd.renderer.render does not need it: it folds the normal pass into the first call.)  Contract when on:
#   * one follow-up per full call: the entry is dropped by the first hit (and by any miss), so a stale hit can only
#     ever be the call immediately after the one that computed the geometry, and the previous call's scratch
#     (~300 MB at 3 M Gaussians) is pinned no longer than that;
#   * identity + version: the previous call's input tensors are kept alive by the entry, so an id / address cannot
#     be recycled; any in-place operation PyTorch knows about bumps `_version` and misses.
import os as _os

_GEOMETRY_CACHE_DEFAULT = _os.environ.get("GSR_GEOMETRY_CACHE", "0") == "1"
_GEOMETRY_CACHE = _GEOMETRY_CACHE_DEFAULT
cache_stats = {"hits": 0, "misses": 0}


def geometry_cache_enabled() -> bool:
    return bool(_GEOMETRY_CACHE)


def set_geometry_cache(enabled) -> None:
    """True / False; None restores the process default (``GSR_GEOMETRY_CACHE``, off when unset)."""
    global _GEOMETRY_CACHE
    _GEOMETRY_CACHE = _GEOMETRY_CACHE_DEFAULT if enabled is None else bool(enabled)
    _tls.cache = None


def _geometry_key(tensors, scalars):
    # "absent" inputs arrive as fresh empty tensors on every call (reference __init__.py:200-210)
    return tuple(("absent",) if t.numel() == 0 else (id(t), t._version, t.data_ptr(), tuple(t.shape))
                 for t in tensors) + tuple(scalars)


def rgba_planes(color: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """``torch.cat((color, alpha), dim=0)`` -- without the copy when the two are the planes of one forward call's
    output buffer (they are whenever they come from this module)."""
    base = color._base
    if (base is not None and alpha._base is base and base.dim() == 3 and base.shape[0] == 4 and base.is_contiguous()
            and color.shape[0] == 3 and alpha.shape[0] == 1 and color.data_ptr() == base.data_ptr()
            and alpha.data_ptr() == base.data_ptr() + 3 * base.stride(0) * base.element_size()):
        return base
    return torch.cat((color, alpha), dim=0)


def last_layout() -> dict:
    """Byte offsets of every scratch sub-array of this thread's most recent forward call, and its pair counts
    (introspection for the parity tests; not part of the reference surface).  The pair counts past the sizing
    read-back stay on the device during a call: asking for them here waits for that call's stream."""
    lay = dict(getattr(_tls, "last_layout", None) or {})
    if lay and "counts" not in lay:
        lay["counts"] = _lib.pair_counts()
        lay["slab_pairs"] = _lib.slab_pairs()
    return lay


def grad_slabs() -> bool:
    """May a forward call that will be differentiated be an inference call (include/gsr.h: GSR_OPT_GRAD_SLABS)?  Not while the
    deterministic backward is requested: that mode sorts one list per tile."""
    return _lib.get_option(_lib.OPT_GRAD_SLABS) != 0 and _lib.get_option(_lib.OPT_BACKWARD_DETERMINISTIC) == 0


import contextlib


@contextlib.contextmanager
def deterministic_backward_guard(slab_forward: bool):
    """``GSR_OPT_BACKWARD_DETERMINISTIC`` must be set BEFORE the forward call it is meant for: the deterministic backward sorts one list
    per tile, and a grad-mode forward made while the option was off may have built its lists per depth slab (``GSR_OPT_GRAD_SLABS``).
    If the option was switched on between such a forward and its backward -- another thread, a test fixture -- the library would refuse
    (``GSR_ERR_INVALID_ARG``); the graph is still differentiable, so this call is made with float atomics instead, with a warning
    (ADVICE round 5).  The choice the forward made rides in the autograd context (``ctx.gsr_slab_forward``)."""
    flip = bool(slab_forward) and _lib.get_option(_lib.OPT_BACKWARD_DETERMINISTIC) != 0
    if flip:
        import warnings
        warnings.warn("GSR_OPT_BACKWARD_DETERMINISTIC was switched on after this graph's forward call, which built its lists per depth slab: "
                      "this backward uses float atomics (set the option before the forward call)", RuntimeWarning, stacklevel=3)
        _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)
    try:
        yield
    finally:
        if flip:
            _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1)


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, *, inference: bool = False
                        ) -> Tuple[int, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor,
                                   torch.Tensor, torch.Tensor, torch.Tensor]:
    """The reference's 19-argument forward (``DGR/rasterize_points.h:18-38``) and its 8-tuple.

    ``inference`` (keyword only, not in the reference): the caller will not differentiate this call, so the library
    may cut the lists into depth slabs and skip what finished tiles no longer need (``GSR_FORWARD_INFERENCE``); the
    images, radii and the returned count are bit-identical either way, the scratch buffers are only good for a second
    blend, not for ``rasterize_gaussians_backward``."""
    return _rasterize(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                      viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                      prefiltered, debug, None, inference)[:8]


def rasterize_gaussians_extra(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                              viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                              prefiltered, debug, extra_colors, *, inference: bool = False):
    """``rasterize_gaussians`` plus a second feature triple per Gaussian composited in the same pass
    (``gsr_forward_extra``): returns the 8-tuple and ``extra_image[3,H,W]``, which equals the colour image of a
    second call with ``colors = extra_colors`` bit for bit.  Not part of the reference surface; used by
    ``autovfx_amd.renderer.render`` for the normal map."""
    if extra_colors is None or extra_colors.dim() != 2 or extra_colors.shape != (means3D.size(0), 3):
        raise RuntimeError("extra_colors must have dimensions (num_points, 3)")
    return _rasterize(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                      viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                      prefiltered, debug, extra_colors, inference)


def _rasterize(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
               viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
               prefiltered, debug, extra_colors, inference=False):
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    device = _require_gpu(means3D, "means3D")
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)

    # The reference zero-fills its outputs (rasterize_points.cu:68-71), which only matters for P == 0:
    # with P > 0 every pixel and every radius is written by the kernels, so the fills are skipped.
    make = (lambda shape, dtype, device: torch.zeros(shape, dtype=dtype, device=device)) if P == 0 else _new
    # colour and alpha are the first three and the fourth plane of ONE buffer, so a caller that wants RGBA
    # (render() does: gaussian_renderer/__init__.py:161) gets it without a copy (rgba_planes below)
    rgba = make((4, H, W), dtype=torch.float32, device=device)
    out_color, out_alpha = rgba[:3], rgba[3:4]
    out_depth = make((1, H, W), dtype=torch.float32, device=device)
    radii = make((P,), dtype=torch.int32, device=device)
    out_extra = make((3, H, W), dtype=torch.float32, device=device) if extra_colors is not None else None
    scratch = _CallScratch(device)
    rendered = 0
    geometry_inputs = (means3D, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos)
    key = None
    if P != 0 and _GEOMETRY_CACHE:
        key = _geometry_key(geometry_inputs, (float(scale_modifier), float(tan_fovx), float(tan_fovy), H, W,
                                              bool(prefiltered), _lib.get_option(_lib.OPT_TILE_CULL), bool(inference),
                                              torch.cuda.current_stream(device).cuda_stream))
        hit = getattr(_tls, "cache", None)
        _tls.cache = None   # whatever happens next, the entry has had its one chance
        if hit is not None and hit["key"] == key and colors.numel() != 0 and sh.numel() == 0 and extra_colors is None:
            cache_stats["hits"] += 1
            return _blend_cached(hit, background, colors, device, H, W, out_color, out_depth, out_alpha)
        del hit
        cache_stats["misses"] += 1
    if P != 0:
        M = int(sh.size(1)) if sh.numel() != 0 else 0
        tensors = [_f32c(n, t, device) for n, t in (
            ("background", background), ("means3D", means3D), ("sh", sh), ("colors", colors), ("opacity", opacity),
            ("scales", scales), ("rotations", rotations), ("cov3D_precomp", cov3D_precomp),
            ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("campos", campos))]
        bg_, m3_, sh_, col_, op_, sc_, rot_, cov_, vm_, pm_, cp_ = tensors
        _tls.call = scratch
        try:
            with torch.cuda.device(device):
                stream = torch.cuda.current_stream(device).cuda_stream
                head = (_GEOM_CB, None, _BINNING_CB, None, _IMAGE_CB, None, P, int(degree), M, _ptr(bg_), W, H,
                        _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(op_), _ptr(sc_), float(scale_modifier), _ptr(rot_),
                        _ptr(cov_), _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx), float(tan_fovy),
                        1 if prefiltered else 0, out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(),
                        radii.data_ptr())
                flags = _lib.FORWARD_INFERENCE if inference else 0
                if extra_colors is None and not flags:
                    rendered = _lib.lib.gsr_forward(*head, 1 if debug else 0, ctypes.c_void_p(stream))
                else:
                    ext_ = None if extra_colors is None else _f32c("extra_colors", extra_colors, device)
                    rendered = _lib.lib.gsr_forward_extra(*head, _ptr(ext_), None if ext_ is None else out_extra.data_ptr(),
                                                          flags, 1 if debug else 0, ctypes.c_void_p(stream))
        finally:
            _tls.call = None
        if rendered < 0:
            raise RuntimeError(f"gsr_forward failed ({rendered}): {_lib.last_error()}")
        # per-thread: the library keeps these per calling thread too, and streams are driven by separate threads
        # (the pair counts are fetched by last_layout() on demand: they would cost a wait for the stream here)
        _tls.last_layout = {"geom": _lib.offsets("geom"), "binning": _lib.offsets("binning"), "image": _lib.offsets("image")}
        if key is not None and extra_colors is None:   # (a fused two-feature call has no second pass to wait for)
            _tls.cache = {"key": key, "inputs": geometry_inputs,
                          "rendered": rendered, "radii": radii, "geom": scratch.buffers["geom"],
                          "binning": scratch.buffers["binning"], "image": scratch.buffers["image"]}
    return (rendered, out_color, out_depth, out_alpha, radii, scratch.buffers["geom"], scratch.buffers["binning"],
            scratch.buffers["image"], out_extra)


class PendingForward:
    """A forward call whose projection and depth sort are queued and whose remaining stages wait for ``finish()``
    (``gsr_forward_begin`` / ``gsr_forward_finish``).  Keeps every tensor the queued kernels read or write alive."""

    def __init__(self, handle, device, stream, scratch, inputs, outputs):
        self._handle, self._device, self._stream = handle, device, stream
        self._scratch, self._inputs, self._outputs = scratch, inputs, outputs

    def ready(self) -> bool:
        """True when ``finish()`` will not wait for the GPU: the call's counters have reached the host."""
        if self._handle is None:
            return True
        r = _lib.lib.gsr_forward_ready(ctypes.c_void_p(self._handle))
        if r < 0:
            raise RuntimeError(f"gsr_forward_ready failed ({r}): {_lib.last_error()}")
        return r != 0

    def finish(self):
        """Queue the rest of the call; returns what ``rasterize_gaussians_extra`` returns.  Call it on the thread
        and with the current stream that ``rasterize_gaussians_begin`` ran on."""
        if self._outputs is None:
            raise RuntimeError("PendingForward.finish() called twice")
        out_color, out_depth, out_alpha, radii, out_extra = self._outputs
        self._outputs = None
        rendered = 0
        if self._handle is not None:
            if torch.cuda.current_stream(self._device).cuda_stream != self._stream:
                _lib.lib.gsr_forward_cancel(ctypes.c_void_p(self._handle))
                self._handle = None
                raise RuntimeError("PendingForward.finish(): the current stream is not the one the call was begun on")
            _tls.call = self._scratch
            try:
                with torch.cuda.device(self._device):
                    rendered = _lib.lib.gsr_forward_finish(ctypes.c_void_p(self._handle))
            finally:
                _tls.call = None
                self._handle = None
            if rendered < 0:
                raise RuntimeError(f"gsr_forward_finish failed ({rendered}): {_lib.last_error()}")
            _tls.last_layout = {"geom": _lib.offsets("geom"), "binning": _lib.offsets("binning"),
                                "image": _lib.offsets("image")}
        b = self._scratch.buffers
        self._inputs = None
        return rendered, out_color, out_depth, out_alpha, radii, b["geom"], b["binning"], b["image"], out_extra

    def __del__(self):
        if getattr(self, "_handle", None) is not None:
            _lib.lib.gsr_forward_cancel(ctypes.c_void_p(self._handle))


def rasterize_gaussians_begin(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                              viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                              prefiltered, debug, extra_colors=None, *, inference: bool = False) -> PendingForward:
    """First half of ``rasterize_gaussians`` (same 19 arguments): everything up to the point where the host has to
    learn the pair count is queued on the current stream and the call returns without waiting.  ``finish()`` on the
    result queues the rest.  One host thread can so keep a frame in flight on each of several streams; results are
    those of the one-shot call, bit for bit.  Not part of the reference surface (its forward is one blocking call,
    ``DGR/rasterize_points.cu:36-119``); used by ``autovfx_amd.frame_parallel``."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    device = _require_gpu(means3D, "means3D")
    P, H, W = int(means3D.size(0)), int(image_height), int(image_width)
    if extra_colors is not None and (extra_colors.dim() != 2 or extra_colors.shape != (P, 3)):
        raise RuntimeError("extra_colors must have dimensions (num_points, 3)")
    make = (lambda shape, dtype, device: torch.zeros(shape, dtype=dtype, device=device)) if P == 0 else _new
    # colour and alpha are the first three and the fourth plane of ONE buffer, so a caller that wants RGBA
    # (render() does: gaussian_renderer/__init__.py:161) gets it without a copy (rgba_planes below)
    rgba = make((4, H, W), dtype=torch.float32, device=device)
    out_color, out_alpha = rgba[:3], rgba[3:4]
    out_depth = make((1, H, W), dtype=torch.float32, device=device)
    radii = make((P,), dtype=torch.int32, device=device)
    out_extra = make((3, H, W), dtype=torch.float32, device=device) if extra_colors is not None else None
    scratch = _CallScratch(device)
    outputs = (out_color, out_depth, out_alpha, radii, out_extra)
    stream = torch.cuda.current_stream(device).cuda_stream
    if P == 0:
        return PendingForward(None, device, stream, scratch, None, outputs)
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    tensors = [_f32c(n, t, device) for n, t in (
        ("background", background), ("means3D", means3D), ("sh", sh), ("colors", colors), ("opacity", opacity),
        ("scales", scales), ("rotations", rotations), ("cov3D_precomp", cov3D_precomp),
        ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("campos", campos))]
    bg_, m3_, sh_, col_, op_, sc_, rot_, cov_, vm_, pm_, cp_ = tensors
    ext_ = _f32c("extra_colors", extra_colors, device) if extra_colors is not None else None
    _tls.call = scratch
    try:
        with torch.cuda.device(device):
            handle = _lib.lib.gsr_forward_begin(
                _GEOM_CB, None, _BINNING_CB, None, _IMAGE_CB, None, P, int(degree), M, _ptr(bg_), W, H,
                _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(op_), _ptr(sc_), float(scale_modifier), _ptr(rot_),
                _ptr(cov_), _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx), float(tan_fovy),
                1 if prefiltered else 0, out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(),
                radii.data_ptr(), _ptr(ext_), None if ext_ is None else out_extra.data_ptr(),
                _lib.FORWARD_INFERENCE if inference else 0, 1 if debug else 0, ctypes.c_void_p(stream))
    finally:
        _tls.call = None
    if not handle:
        raise RuntimeError(f"gsr_forward_begin failed: {_lib.last_error()}")
    return PendingForward(handle, device, stream, scratch, (tensors, ext_), outputs)


def _raw_call(begin, background, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, scale_modifier,
              viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree, campos, prefiltered, debug,
              want_normal, inference):
    """Common part of ``rasterize_gaussians_raw`` / ``rasterize_gaussians_raw_begin``."""
    if xyz.dim() != 2 or xyz.size(1) != 3:
        raise RuntimeError("xyz must have dimensions (num_points, 3)")
    device = _require_gpu(xyz, "xyz")
    P, H, W = int(xyz.size(0)), int(image_height), int(image_width)
    if features_dc.dim() != 3 or tuple(features_dc.shape) != (P, 1, 3):
        raise RuntimeError("features_dc must have dimensions (num_points, 1, 3)")
    if features_rest.dim() != 3 or features_rest.size(0) != P or features_rest.size(2) != 3:
        raise RuntimeError("features_rest must have dimensions (num_points, M - 1, 3)")
    if tuple(log_scales.shape) != (P, 3) or tuple(rotations.shape) != (P, 4) or opacity_logits.numel() != P:
        raise RuntimeError("log_scales / rotations / opacity_logits must have dimensions (num_points, 3 / 4 / 1)")
    M = 1 + int(features_rest.size(1))
    make = (lambda shape, dtype, device: torch.zeros(shape, dtype=dtype, device=device)) if P == 0 else _new
    rgba = make((4, H, W), dtype=torch.float32, device=device)
    out_color, out_alpha = rgba[:3], rgba[3:4]
    out_depth = make((1, H, W), dtype=torch.float32, device=device)
    radii = make((P,), dtype=torch.int32, device=device)
    out_normal = make((3, H, W), dtype=torch.float32, device=device) if want_normal else None
    scratch = _CallScratch(device)
    outputs = (out_color, out_depth, out_alpha, radii, out_normal)
    stream = torch.cuda.current_stream(device).cuda_stream
    if P == 0:
        return None, device, stream, scratch, None, outputs
    tensors = [_f32c(n, t, device) for n, t in (
        ("background", background), ("xyz", xyz), ("log_scales", log_scales), ("rotations", rotations),
        ("opacity_logits", opacity_logits), ("features_dc", features_dc), ("features_rest", features_rest),
        ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("campos", campos))]
    bg_, xyz_, ls_, rot_, op_, dc_, rest_, vm_, pm_, cp_ = tensors
    raw = _lib.RawParams(_ptr(xyz_), _ptr(ls_), _ptr(rot_), _ptr(op_), _ptr(dc_), _ptr(rest_))
    _tls.call = scratch
    try:
        with torch.cuda.device(device):
            fn = _lib.lib.gsr_forward_raw_begin if begin else _lib.lib.gsr_forward_raw
            rc = fn(_GEOM_CB, None, _BINNING_CB, None, _IMAGE_CB, None, P, int(degree), M, _ptr(bg_), W, H, ctypes.byref(raw),
                    float(scale_modifier), _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx), float(tan_fovy),
                    1 if prefiltered else 0, out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(), radii.data_ptr(),
                    None if out_normal is None else out_normal.data_ptr(), _lib.FORWARD_INFERENCE if inference else 0,
                    1 if debug else 0, ctypes.c_void_p(stream))
    finally:
        _tls.call = None
    return rc, device, stream, scratch, tensors, outputs


def rasterize_gaussians_raw(background, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest,
                            scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree,
                            campos, prefiltered, debug, *, want_normal: bool = True, inference: bool = True):
    """The forward pass straight from a model's RAW parameter tensors (``gsr_forward_raw``): ``_xyz``, ``_scaling`` (log),
    ``_rotation`` (unnormalised), ``_opacity`` (logit), ``_features_dc [P,1,3]``, ``_features_rest [P,M-1,3]`` as
    ``GaussianModel`` stores them (``scene/gaussian_model.py:95-128``).  ``exp`` / ``sigmoid`` / ``F.normalize``, the
    ``cat(dc, rest)`` and -- with ``want_normal`` -- ``get_normal(dir) * 0.5 + 0.5`` happen inside the kernels, rounded as
    PyTorch rounds them on this GPU: the result is bit-identical to activating in PyTorch and calling
    ``rasterize_gaussians_extra``.  Returns that function's 9-tuple (the last entry is the normal image or None).  Not
    part of the reference surface; ``autovfx_amd.renderer.render`` uses it for any model that exposes the six tensors."""
    rc, device, stream, scratch, tensors, outputs = _raw_call(
        False, background, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, scale_modifier, viewmatrix,
        projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree, campos, prefiltered, debug, want_normal, inference)
    out_color, out_depth, out_alpha, radii, out_normal = outputs
    rendered = 0
    if rc is not None:
        if rc < 0:
            raise RuntimeError(f"gsr_forward_raw failed ({rc}): {_lib.last_error()}")
        rendered = rc
        _tls.last_layout = {"geom": _lib.offsets("geom"), "binning": _lib.offsets("binning"), "image": _lib.offsets("image")}
    b = scratch.buffers
    return rendered, out_color, out_depth, out_alpha, radii, b["geom"], b["binning"], b["image"], out_normal


def rasterize_gaussians_raw_begin(background, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest,
                                  scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width,
                                  degree, campos, prefiltered, debug, *, want_normal: bool = True,
                                  inference: bool = True) -> PendingForward:
    """``rasterize_gaussians_raw`` in two halves (``gsr_forward_raw_begin``): see ``rasterize_gaussians_begin``."""
    handle, device, stream, scratch, tensors, outputs = _raw_call(
        True, background, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, scale_modifier, viewmatrix,
        projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree, campos, prefiltered, debug, want_normal, inference)
    if tensors is None:   # P == 0
        return PendingForward(None, device, stream, scratch, None, outputs)
    if not handle:
        raise RuntimeError(f"gsr_forward_raw_begin failed: {_lib.last_error()}")
    return PendingForward(handle, device, stream, scratch, (tensors, None), outputs)


def rasterize_gaussians_raw_backward(background, xyz, log_scales, rotations, opacity_logits, features_dc, features_rest, radii,
                                     scale_modifier, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                     dL_dout_alpha, dL_dout_normal, degree, campos, geomBuffer, R, binningBuffer, imageBuffer,
                                     out_alpha, debug):
    """Gradients of a full ``rasterize_gaussians_raw`` call with respect to the model's raw tensors (``gsr_backward_raw``):
    returns ``(dL_dmeans2D, dL_dxyz, dL_dlog_scales, dL_drotations, dL_dopacity_logits, dL_dfeatures_dc, dL_dfeatures_rest)``.
    ``dL_dout_depth`` / ``dL_dout_alpha`` / ``dL_dout_normal`` may be None (no gradient for that image)."""
    device = _require_gpu(xyz, "xyz")
    P = int(xyz.size(0))
    H, W = int(out_alpha.size(-2)), int(out_alpha.size(-1))
    M = 1 + int(features_rest.size(1))
    z = lambda *shape: _new(shape, torch.float32, device)
    g2d, gxyz, gls, grot, gop = z(P, 3), z(P, 3), z(P, 3), z(P, 4), z(*opacity_logits.shape)
    gdc, grest = z(P, 1, 3), z(P, M - 1, 3)
    if P != 0:
        f = lambda n, t: _f32c(n, t, device)
        fn_ = lambda n, t: None if t is None else f(n, t)
        if dL_dout_color is None:
            dL_dout_color = torch.zeros((3, H, W), dtype=torch.float32, device=device)
        if (dL_dout_depth is None) != (dL_dout_alpha is None):
            dL_dout_depth = torch.zeros((1, H, W), dtype=torch.float32, device=device) if dL_dout_depth is None else dL_dout_depth
            dL_dout_alpha = torch.zeros((1, H, W), dtype=torch.float32, device=device) if dL_dout_alpha is None else dL_dout_alpha
        tensors = [f(n, t) for n, t in (("background", background), ("xyz", xyz), ("log_scales", log_scales), ("rotations", rotations),
                                        ("opacity_logits", opacity_logits), ("features_dc", features_dc), ("features_rest", features_rest),
                                        ("viewmatrix", viewmatrix), ("projmatrix", projmatrix), ("campos", campos),
                                        ("out_alpha", out_alpha), ("dL_dout_color", dL_dout_color))]
        bg_, xyz_, ls_, rot_, op_, dc_, rest_, vm_, pm_, cp_, oa_, gc_ = tensors
        gd_, ga_, gn_ = fn_("dL_dout_depth", dL_dout_depth), fn_("dL_dout_alpha", dL_dout_alpha), fn_("dL_dout_normal", dL_dout_normal)
        radii_ = radii.contiguous()
        if radii_.dtype != torch.int32:
            raise RuntimeError(f"radii: expected an int32 tensor, got {radii_.dtype}")
        raw = _lib.RawParams(_ptr(xyz_), _ptr(ls_), _ptr(rot_), _ptr(op_), _ptr(dc_), _ptr(rest_))
        accum = _new((P, 16), torch.float32, device)   # cleared by the library
        with torch.cuda.device(device):
            rc = _lib.lib.gsr_backward_raw(
                P, int(degree), M, int(R), _ptr(bg_), W, H, ctypes.byref(raw), float(scale_modifier), _ptr(vm_), _ptr(pm_), _ptr(cp_),
                float(tan_fovx), float(tan_fovy), _ptr(radii_), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(oa_),
                _ptr(gc_), _ptr(gd_), _ptr(ga_), _ptr(gn_), g2d.data_ptr(), gxyz.data_ptr(), gls.data_ptr(), grot.data_ptr(),
                gop.data_ptr(), gdc.data_ptr(), grest.data_ptr() if M > 1 else None, accum.data_ptr(), 1 if debug else 0,
                ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gsr_backward_raw failed ({rc}): {_lib.last_error()}")
    return g2d, gxyz, gls, grot, gop, gdc, grest


def _blend_cached(hit, background, colors, device, H, W, out_color, out_depth, out_alpha):
    """Second pass over cached geometry: one blend launch over the first pass's lists (gsr_blend)."""
    geom, binning, image = hit["geom"], hit["binning"], hit["image"]
    bg_, col_ = _f32c("background", background, device), _f32c("colors", colors, device)
    with torch.cuda.device(device):
        rc = _lib.lib.gsr_blend(geom.data_ptr(), binning.data_ptr(), image.data_ptr(), W, H, col_.data_ptr(), bg_.data_ptr(),
                                out_color.data_ptr(), out_depth.data_ptr(), out_alpha.data_ptr(),
                                ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gsr_blend failed ({rc}): {_lib.last_error()}")
    return hit["rendered"], out_color, out_depth, out_alpha, hit["radii"].clone(), geom, binning, image, None


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_depth,
                                 dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, out_alpha,
                                 debug, *, skip_unused: bool = False):
    """Gradients of one forward call: the 24-argument backward of the reference
    (``DGR/rasterize_points.h:40-65``, ``DGR/rasterize_points.cu:121-209``), same order in, same 8-tuple out
    ``(dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)``.
    ``skip_unused`` (keyword only, not in the reference): gradients of inputs that were not given -- ``dL_dcolors`` with SH
    colours, ``dL_dcov3D`` with scales and rotations -- and the two intermediates the reference computes but never returns
    (conic, depth) are not written at all (56 bytes per Gaussian less); the skipped entries of the tuple are None."""
    device = _require_gpu(means3D, "means3D")
    P = int(means3D.size(0))
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    M = int(sh.size(1)) if sh.numel() != 0 else 0
    # P == 0: nothing runs, the (empty) results are trivially defined; otherwise gsr_backward writes every element
    z = lambda *shape: _new(shape, torch.float32, device)
    want_colors = not (skip_unused and colors.numel() == 0)
    want_cov = not (skip_unused and cov3D_precomp.numel() == 0)
    dL_dmeans3D, dL_dmeans2D = z(P, 3), z(P, 3)
    dL_dcolors = z(P, 3) if want_colors else None
    dL_ddepths, dL_dconic = (None, None) if skip_unused else (z(P, 1), z(P, 2, 2))   # intermediates
    dL_dopacity = z(P, 1)
    dL_dcov3D = z(P, 6) if want_cov else None
    dL_dsh, dL_dscales, dL_drotations = z(P, M, 3), z(P, 3), z(P, 4)
    if P != 0:
        f = lambda n, t: _f32c(n, t, device)
        # (beyond the reference: dL_dout_depth / dL_dout_alpha may be None = all zeros -- a loss that only reads the colour image;
        # the per-pixel pass then skips their terms.  One of them alone is completed with zeros.)
        if (dL_dout_depth is None) != (dL_dout_alpha is None):
            like = (H, W)
            dL_dout_depth = torch.zeros((1, *like), dtype=torch.float32, device=device) if dL_dout_depth is None else dL_dout_depth
            dL_dout_alpha = torch.zeros((1, *like), dtype=torch.float32, device=device) if dL_dout_alpha is None else dL_dout_alpha
        fn_ = lambda n, t: None if t is None else f(n, t)
        bg_, m3_, sh_, col_, sc_, rot_, cov_, vm_, pm_, cp_, oa_, gc_, gd_, ga_ = (
            f("background", background), f("means3D", means3D), f("sh", sh), f("colors", colors), f("scales", scales),
            f("rotations", rotations), f("cov3D_precomp", cov3D_precomp), f("viewmatrix", viewmatrix),
            f("projmatrix", projmatrix), f("campos", campos), f("out_alpha", out_alpha),
            f("dL_dout_color", dL_dout_color), fn_("dL_dout_depth", dL_dout_depth), fn_("dL_dout_alpha", dL_dout_alpha))
        radii_ = radii.contiguous()
        if radii_.dtype != torch.int32:
            raise RuntimeError(f"radii: expected an int32 tensor, got {radii_.dtype}")
        accum = _new((P, 16), torch.float32, device)   # cleared by the library
        with torch.cuda.device(device):
            rc = _lib.lib.gsr_backward(
                P, int(degree), M, int(R), _ptr(bg_), W, H, _ptr(m3_), _ptr(sh_), _ptr(col_), _ptr(sc_),
                float(scale_modifier), _ptr(rot_), _ptr(cov_), _ptr(vm_), _ptr(pm_), _ptr(cp_), float(tan_fovx),
                float(tan_fovy), _ptr(radii_), _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imageBuffer), _ptr(oa_),
                _ptr(gc_), _ptr(gd_), _ptr(ga_), dL_dmeans2D.data_ptr(), _ptr(dL_dconic), dL_dopacity.data_ptr(),
                _ptr(dL_dcolors), _ptr(dL_ddepths), dL_dmeans3D.data_ptr(), _ptr(dL_dcov3D),
                dL_dsh.data_ptr() if M else None, dL_dscales.data_ptr(), dL_drotations.data_ptr(), accum.data_ptr(),
                1 if debug else 0,
                ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gsr_backward failed ({rc}): {_lib.last_error()}")
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix) -> torch.Tensor:
    device = _require_gpu(means3D, "means3D")
    P = int(means3D.size(0))
    present = torch.zeros((P,), dtype=torch.bool, device=device)
    if P != 0:
        m3_ = _f32c("means3D", means3D, device)
        vm_ = _f32c("viewmatrix", viewmatrix, device)
        pm_ = _f32c("projmatrix", projmatrix, device)
        with torch.cuda.device(device):
            rc = _lib.lib.gsr_mark_visible(P, _ptr(m3_), _ptr(vm_), _ptr(pm_), present.data_ptr(),
                                           ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gsr_mark_visible failed ({rc}): {_lib.last_error()}")
    return present
