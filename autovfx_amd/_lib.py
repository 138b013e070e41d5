"""ctypes loader for libgsr_hip.so, the C-ABI rasterizer declared in include/gsr.h.

There is deliberately no fallback: if the shared library is missing or does not export the full
ABI, importing this module raises, and so does every package that depends on it
(``diff_gaussian_rasterization``).  Build the library with ``python -m autovfx_amd.build``.
"""
from __future__ import annotations

import ctypes
import os

# PyTorch first, always: its wheel carries its own libamdhip64 / libhsa-runtime64, and the library below asks the loader for
# "libamdhip64.so.7" by name.  Loaded after torch it binds to the runtime torch already brought in -- one HIP runtime in the
# process, so torch's stream handles and allocations mean the same thing on both sides.  Loaded BEFORE torch it would pull in the
# system's runtime, torch's own would follow, and the mixture fails at the first call ("no ROCm-capable device is detected").
import torch  # noqa: F401  (import order matters, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_LIB", os.path.join(_HERE, "lib", "libgsr_hip.so"))

ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)

# enum mirrors of include/gsr.h
GEOM_SLOTS = ("raster", "rgb", "splat_bins", "internal_radii", "depth_order", "point_offsets", "listed", "view_normals")
BIN_SLOTS = ("point_list", "tile_keys")
IMG_SLOTS = ("ranges", "n_contrib")
STAGES = ("preprocess", "depth_sort", "scan", "duplicate", "tile_sort", "ranges", "blend", "colour")
ABI_VERSION = 14


class PngFileInfo(ctypes.Structure):
    """``GsrPngFileInfo`` (gsr.h)."""
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("channels", ctypes.c_int), ("scanline_bytes", ctypes.c_size_t)]


class ExrFileInfo(ctypes.Structure):
    """``GsrExrFileInfo`` (gsr.h)."""
    _fields_ = [("width", ctypes.c_int), ("height", ctypes.c_int), ("bytes_per_line", ctypes.c_int), ("lines_per_block", ctypes.c_int),
                ("channel_at", ctypes.c_int), ("channel_bytes", ctypes.c_int), ("channel_is_half", ctypes.c_int), ("compression", ctypes.c_int),
                ("n_blocks", ctypes.c_int), ("blocks_bytes", ctypes.c_size_t),
                ("channel", ctypes.c_char * 32)]


class PngUnfilterJob(ctypes.Structure):
    """``GsrPngUnfilterJob`` (gsr.h)."""
    _fields_ = [("scanlines", ctypes.c_void_p), ("width", ctypes.c_int), ("height", ctypes.c_int), ("channels", ctypes.c_int),
                ("out_rgba", ctypes.c_void_p), ("scratch", ctypes.c_void_p)]
MAX_SLABS = 8
FORWARD_INFERENCE = 1

# every symbol include/gsr.h declares
SYMBOLS = ("gsr_forward", "gsr_mark_visible", "gsr_backward", "gsr_last_geom_offsets", "gsr_last_binning_offsets",
           "gsr_last_image_offsets", "gsr_set_stage_timing", "gsr_get_stage_times", "gsr_last_error",
           "gsr_abi_version", "gsr_target_arch", "gsr_set_option", "gsr_get_option", "gsr_pack_rgba8", "gsr_png_size", "gsr_png_room", "gsr_png_encode", "gsr_frame_files", "gsr_png_deflate_max_size", "gsr_png_deflate_room", "gsr_png_deflate_scratch", "gsr_png_encode_deflate", "gsr_frame_files_deflate", "gsr_resize_rgba8_bilinear", "gsr_resize_f32_nearest", "gsr_png_unfilter_scratch", "gsr_png_unfilter", "gsr_png_unfilter_batch", "gsr_exr_unpack_channel", "gsr_upload", "gsr_png_file_probe", "gsr_png_file_inflate", "gsr_exr_file_probe", "gsr_exr_file_inflate", "gsr_selftest_inflate_host", "gsr_exr_file_pack", "gsr_inflate_zlib_blocks", "gsr_last_pair_counts", "gsr_blend", "gsr_composite",
           "gsr_radix_scratch_bytes", "gsr_radix_sort_pairs", "gsr_selftest_exp", "gsr_view_normals", "gsr_normal_maps", "gsr_forward_extra", "gsr_get_call_times",
           "gsr_forward_begin", "gsr_forward_finish", "gsr_forward_ready", "gsr_forward_cancel", "gsr_last_slab_pairs", "gsr_plan_slabs", "gsr_selftest_lds_atomic_order", "gsr_get_backward_times", "gsr_place_object",
           "gsr_forward_raw", "gsr_forward_raw_begin", "gsr_backward_raw", "gsr_place_object_subset")
OPT_TILE_CULL = 0
OPT_SLABS = 1
OPT_SLAB_FIRST = 2
OPT_DEFER_COLOUR = 3
OPT_SLAB_MIN_REST = 4
OPT_RADIX_RANK = 5           # 0 ballots, 1 verified LDS adds, 2 (default) those where the per-device self-test passed, 3 test hook
OPT_RADIX_RANK_ACTIVE = 6    # read-only: what the current device uses (1 verified LDS adds, 0 ballots)
OPT_BLEND_ORDER = 8          # 1 (default): large images are blended longest tile list first inside each XCD's band
OPT_DEPTH_DROP = 7           # 1 (default): Gaussians that emit nothing leave the depth sort in its first pass
OPT_BACKWARD_DETERMINISTIC = 10  # 1: per-Gaussian gradient sums in a fixed order (same bits on every run); default 0: float atomics
OPT_GRAD_SLABS = 11              # 1 (default): a grad-mode forward may be an inference call (depth slabs, deferred colours); the backward walks the slabs
OPT_RADIX_RANK_FALLBACKS = 9 # read-only: tiles on the current device whose LDS-add ranks failed the order check (re-ranked with ballots)


class RawParams(ctypes.Structure):
    """``gsr_raw_params`` (include/gsr.h): device pointers of a model's six raw parameter tensors."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("xyz", "log_scales", "rotations", "opacity_logits", "features_dc", "features_rest")]


class GsrLibraryError(ImportError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise GsrLibraryError(
            f"{LIB_PATH} not found: the HIP rasterizer is not built. Run `python -m autovfx_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64.so missing
        raise GsrLibraryError(f"could not load {LIB_PATH}: {e}") from e
    missing = [s for s in SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise GsrLibraryError(f"{LIB_PATH} does not export {missing}; rebuild it")

    c_f = ctypes.c_void_p  # device float* / int* travel as integers (tensor.data_ptr())
    lib.gsr_forward.restype = ctypes.c_int
    lib.gsr_forward.argtypes = [
        ALLOC_FN, ctypes.c_void_p, ALLOC_FN, ctypes.c_void_p, ALLOC_FN, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int,            # P D M
        c_f, ctypes.c_int, ctypes.c_int,                     # background width height
        c_f, c_f, c_f, c_f, c_f, ctypes.c_float, c_f, c_f,   # means3D shs colors opacities scales mod rotations cov3D
        c_f, c_f, c_f, ctypes.c_float, ctypes.c_float, ctypes.c_int,  # view proj campos tanx tany prefiltered
        c_f, c_f, c_f, c_f, ctypes.c_int, ctypes.c_void_p]   # out_color out_depth out_alpha radii debug stream
    lib.gsr_forward_extra.restype = ctypes.c_int
    lib.gsr_forward_extra.argtypes = lib.gsr_forward.argtypes[:-2] + [c_f, c_f, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]
    lib.gsr_forward_begin.restype = ctypes.c_void_p
    lib.gsr_forward_begin.argtypes = lib.gsr_forward_extra.argtypes
    lib.gsr_forward_raw.restype = ctypes.c_int
    lib.gsr_forward_raw.argtypes = [
        ALLOC_FN, ctypes.c_void_p, ALLOC_FN, ctypes.c_void_p, ALLOC_FN, ctypes.c_void_p,
        ctypes.c_int, ctypes.c_int, ctypes.c_int,            # P D M
        c_f, ctypes.c_int, ctypes.c_int,                     # background width height
        ctypes.POINTER(RawParams), ctypes.c_float,           # raw scale_modifier
        c_f, c_f, c_f, ctypes.c_float, ctypes.c_float, ctypes.c_int,  # view proj campos tanx tany prefiltered
        c_f, c_f, c_f, c_f, c_f, ctypes.c_uint, ctypes.c_int, ctypes.c_void_p]   # color depth alpha radii normal flags debug stream
    lib.gsr_forward_raw_begin.restype = ctypes.c_void_p
    lib.gsr_forward_raw_begin.argtypes = lib.gsr_forward_raw.argtypes
    lib.gsr_forward_finish.restype = ctypes.c_int
    lib.gsr_forward_finish.argtypes = [ctypes.c_void_p]
    lib.gsr_forward_ready.restype = ctypes.c_int
    lib.gsr_forward_ready.argtypes = [ctypes.c_void_p]
    lib.gsr_forward_cancel.restype = None
    lib.gsr_forward_cancel.argtypes = [ctypes.c_void_p]
    lib.gsr_mark_visible.restype = ctypes.c_int
    lib.gsr_mark_visible.argtypes = [ctypes.c_int, c_f, c_f, c_f, c_f, ctypes.c_void_p]
    lib.gsr_composite.restype = ctypes.c_int
    lib.gsr_composite.argtypes = [ctypes.c_int, ctypes.c_int] + [c_f] * 12 + [ctypes.c_void_p]
    lib.gsr_blend.restype = ctypes.c_int
    lib.gsr_blend.argtypes = [c_f, c_f, c_f, ctypes.c_int, ctypes.c_int] + [c_f] * 5 + [ctypes.c_void_p]
    lib.gsr_last_slab_pairs.restype = ctypes.c_int
    lib.gsr_plan_slabs.restype = ctypes.c_int
    lib.gsr_plan_slabs.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint32 * MAX_SLABS)]
    lib.gsr_last_slab_pairs.argtypes = [ctypes.POINTER(ctypes.c_uint32 * MAX_SLABS)]
    lib.gsr_last_pair_counts.restype = ctypes.c_int
    lib.gsr_last_pair_counts.argtypes = [ctypes.POINTER(ctypes.c_uint32 * 2)]
    lib.gsr_radix_scratch_bytes.restype = ctypes.c_size_t
    lib.gsr_radix_scratch_bytes.argtypes = [ctypes.c_uint32, ctypes.c_int]
    lib.gsr_radix_sort_pairs.restype = ctypes.c_int
    lib.gsr_radix_sort_pairs.argtypes = [ctypes.c_uint32, ctypes.c_int, c_f, c_f, c_f, c_f, ctypes.c_int, c_f,
                                         ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
    lib.gsr_view_normals.restype = ctypes.c_int
    lib.gsr_view_normals.argtypes = [ctypes.c_int, c_f, c_f, c_f, c_f, ctypes.c_void_p]
    lib.gsr_normal_maps.restype = ctypes.c_int
    lib.gsr_normal_maps.argtypes = [ctypes.c_int, ctypes.c_int, c_f, c_f, c_f] + [ctypes.c_float] * 4 + [c_f, c_f, ctypes.c_void_p]
    lib.gsr_place_object.restype = ctypes.c_int
    lib.gsr_place_object.argtypes = [ctypes.c_int, c_f, c_f, c_f, c_f, c_f, ctypes.c_int, ctypes.POINTER(ctypes.c_float * 21),
                                     c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p]
    lib.gsr_place_object_subset.restype = ctypes.c_int
    lib.gsr_place_object_subset.argtypes = [ctypes.c_int, c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_int, ctypes.POINTER(ctypes.c_float * 21),
                                            c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p]
    lib.gsr_selftest_exp.restype = ctypes.c_int
    lib.gsr_selftest_lds_atomic_order.restype = ctypes.c_int
    lib.gsr_selftest_lds_atomic_order.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    lib.gsr_selftest_exp.argtypes = [ctypes.c_uint32, ctypes.c_uint32, c_f, ctypes.c_void_p]
    lib.gsr_pack_rgba8.restype = ctypes.c_int
    lib.gsr_pack_rgba8.argtypes = [c_f, c_f, c_f, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.gsr_png_size.restype = ctypes.c_size_t
    lib.gsr_png_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.gsr_png_room.restype = ctypes.c_size_t
    lib.gsr_png_room.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.gsr_png_encode.restype = ctypes.c_int
    lib.gsr_png_encode.argtypes = [c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f, ctypes.c_void_p]
    lib.gsr_png_deflate_max_size.restype = ctypes.c_size_t
    lib.gsr_png_deflate_max_size.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.gsr_png_deflate_room.restype = ctypes.c_size_t
    lib.gsr_png_deflate_room.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.gsr_png_deflate_scratch.restype = ctypes.c_size_t
    lib.gsr_png_deflate_scratch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.gsr_png_encode_deflate.restype = ctypes.c_int
    lib.gsr_png_encode_deflate.argtypes = [c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f, c_f, c_f, ctypes.c_void_p]
    lib.gsr_frame_files_deflate.restype = ctypes.c_int
    lib.gsr_frame_files_deflate.argtypes = [c_f, c_f, c_f, c_f, ctypes.c_float, c_f, ctypes.c_int, ctypes.c_int, c_f, c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p]
    lib.gsr_frame_files.restype = ctypes.c_int
    lib.gsr_frame_files.argtypes = [c_f, c_f, c_f, c_f, ctypes.c_float, c_f, ctypes.c_int, ctypes.c_int, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p]
    lib.gsr_png_unfilter_scratch.restype = ctypes.c_size_t
    lib.gsr_png_unfilter_scratch.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.gsr_png_unfilter.restype = ctypes.c_int
    lib.gsr_png_unfilter.argtypes = [c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f, c_f, ctypes.c_void_p]
    lib.gsr_png_unfilter_batch.restype = ctypes.c_int
    lib.gsr_png_unfilter_batch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.gsr_exr_unpack_channel.restype = ctypes.c_int
    lib.gsr_exr_unpack_channel.argtypes = [c_f, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f, ctypes.c_void_p]
    lib.gsr_png_file_probe.restype = ctypes.c_int
    lib.gsr_png_file_probe.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.gsr_png_file_inflate.restype = ctypes.c_int
    lib.gsr_png_file_inflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.gsr_exr_file_probe.restype = ctypes.c_int
    lib.gsr_exr_file_probe.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p]
    lib.gsr_exr_file_inflate.restype = ctypes.c_int
    lib.gsr_exr_file_inflate.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.gsr_exr_file_pack.restype = ctypes.c_int
    lib.gsr_exr_file_pack.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    lib.gsr_inflate_zlib_blocks.restype = ctypes.c_int
    lib.gsr_inflate_zlib_blocks.argtypes = [c_f, c_f, c_f, ctypes.c_int, c_f, c_f, ctypes.c_void_p]
    lib.gsr_selftest_inflate_host.restype = ctypes.c_int
    lib.gsr_selftest_inflate_host.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    lib.gsr_upload.restype = ctypes.c_int
    lib.gsr_upload.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.gsr_resize_rgba8_bilinear.restype = ctypes.c_int
    lib.gsr_resize_rgba8_bilinear.argtypes = [c_f, ctypes.c_int, ctypes.c_int, c_f, ctypes.c_int, ctypes.c_int, c_f, ctypes.c_void_p]
    lib.gsr_resize_f32_nearest.restype = ctypes.c_int
    lib.gsr_resize_f32_nearest.argtypes = [c_f, ctypes.c_int, ctypes.c_int, c_f, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.gsr_backward.restype = ctypes.c_int
    lib.gsr_backward.argtypes = [
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f, ctypes.c_int, ctypes.c_int,  # P D M R bg W H
        c_f, c_f, c_f, c_f, ctypes.c_float, c_f, c_f,          # means3D shs colors scales mod rotations cov3D
        c_f, c_f, c_f, ctypes.c_float, ctypes.c_float,         # view proj campos tanx tany
        c_f, c_f, c_f, c_f,                                    # radii geom binning image
        c_f, c_f, c_f, c_f,                                    # accum_alphas dL_dpix dL_dpix_depth dL_dpix_alpha
        c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f,      # dL_dmean2D conic opacity color depth mean3D cov3D sh scale rot
        c_f, ctypes.c_int, ctypes.c_void_p]                    # accum_scratch debug stream
    lib.gsr_backward_raw.restype = ctypes.c_int
    lib.gsr_backward_raw.argtypes = [
        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_f, ctypes.c_int, ctypes.c_int,  # P D M R bg W H
        ctypes.POINTER(RawParams), ctypes.c_float,             # raw scale_modifier
        c_f, c_f, c_f, ctypes.c_float, ctypes.c_float,         # view proj campos tanx tany
        c_f, c_f, c_f, c_f,                                    # radii geom binning image
        c_f, c_f, c_f, c_f, c_f,                               # accum_alphas dL_dpix dL_dpix_depth dL_dpix_alpha dL_dpix_normal
        c_f, c_f, c_f, c_f, c_f, c_f, c_f,                     # dL_dmean2D xyz log_scales rotations opacity dc rest
        c_f, ctypes.c_int, ctypes.c_void_p]                    # accum_scratch debug stream
    for name, n in (("gsr_last_geom_offsets", len(GEOM_SLOTS)), ("gsr_last_binning_offsets", len(BIN_SLOTS)),
                    ("gsr_last_image_offsets", len(IMG_SLOTS))):
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(ctypes.c_size_t * n)]
    lib.gsr_set_option.restype = ctypes.c_int
    lib.gsr_set_option.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.gsr_get_option.restype = ctypes.c_int
    lib.gsr_get_option.argtypes = [ctypes.c_int]
    lib.gsr_set_stage_timing.restype = None
    lib.gsr_set_stage_timing.argtypes = [ctypes.c_int]
    lib.gsr_get_call_times.restype = ctypes.c_int
    lib.gsr_get_call_times.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_int]
    lib.gsr_get_backward_times.restype = ctypes.c_int
    lib.gsr_get_backward_times.argtypes = [ctypes.POINTER(ctypes.c_float * 2)]
    lib.gsr_get_stage_times.restype = ctypes.c_int
    lib.gsr_get_stage_times.argtypes = [ctypes.POINTER(ctypes.c_float * len(STAGES))]
    lib.gsr_last_error.restype = ctypes.c_char_p
    lib.gsr_last_error.argtypes = []
    lib.gsr_abi_version.restype = ctypes.c_int
    lib.gsr_abi_version.argtypes = []
    lib.gsr_target_arch.restype = ctypes.c_char_p
    lib.gsr_target_arch.argtypes = []
    if lib.gsr_abi_version() != ABI_VERSION:
        raise GsrLibraryError(f"{LIB_PATH} has ABI {lib.gsr_abi_version()}, this binding expects {ABI_VERSION}")
    return lib


lib = _load()

# A/B from the shell: GSR_RADIX_RANK=0 forces the ballot rank of the radix sort, 1 the verified LDS adds, 2 = those where the
# per-device self-test passed, 3 = with an injected inversion (test hook); unset = the library default (2).
if os.environ.get("GSR_RADIX_RANK", "") in ("0", "1", "2", "3"):
    lib.gsr_set_option(5, int(os.environ["GSR_RADIX_RANK"]))
if os.environ.get("GSR_BLEND_ORDER", "") in ("0", "1"):
    lib.gsr_set_option(OPT_BLEND_ORDER, int(os.environ["GSR_BLEND_ORDER"]))
if os.environ.get("GSR_DEPTH_DROP", "") in ("0", "1"):
    lib.gsr_set_option(OPT_DEPTH_DROP, int(os.environ["GSR_DEPTH_DROP"]))
if os.environ.get("GSR_GRAD_SLABS", "") in ("0", "1"):
    lib.gsr_set_option(OPT_GRAD_SLABS, int(os.environ["GSR_GRAD_SLABS"]))
if os.environ.get("GSR_BACKWARD_DETERMINISTIC", "") in ("0", "1"):   # same gradient bits on every run (include/gsr.h)
    lib.gsr_set_option(OPT_BACKWARD_DETERMINISTIC, int(os.environ["GSR_BACKWARD_DETERMINISTIC"]))


def last_error() -> str:
    return lib.gsr_last_error().decode("utf-8", "replace")


def offsets(kind: str) -> dict:
    names, fn = {"geom": (GEOM_SLOTS, lib.gsr_last_geom_offsets), "binning": (BIN_SLOTS, lib.gsr_last_binning_offsets),
                 "image": (IMG_SLOTS, lib.gsr_last_image_offsets)}[kind]
    arr = (ctypes.c_size_t * len(names))()
    if fn(ctypes.byref(arr)) != 0:
        raise RuntimeError(last_error())
    return dict(zip(names, (int(v) for v in arr)))


def pair_counts() -> dict:
    arr = (ctypes.c_uint32 * 2)()
    if lib.gsr_last_pair_counts(ctypes.byref(arr)) != 0:
        raise RuntimeError(last_error())
    return {"num_rendered": int(arr[0]), "live_pairs": int(arr[1])}


def slab_pairs() -> list:
    """Pairs in the list of each depth slab of this thread's last forward call (waits for that call's stream)."""
    arr = (ctypes.c_uint32 * MAX_SLABS)()
    n = lib.gsr_last_slab_pairs(ctypes.byref(arr))
    if n < 0:
        raise RuntimeError(last_error())
    return [int(arr[i]) for i in range(n)]


def plan_slabs(live_pairs: int, width: int, height: int) -> list:
    """Inclusive pair offsets at which an inference call with that many live pairs would cut its depth slabs under the
    current options ([] = one slab).  Host logic only."""
    arr = (ctypes.c_uint32 * MAX_SLABS)()
    n = lib.gsr_plan_slabs(int(live_pairs), int(width), int(height), ctypes.byref(arr))
    if n < 0:
        raise RuntimeError(last_error())
    return [int(arr[i]) for i in range(n - 1)]


def set_option(option: int, value: int) -> None:
    if lib.gsr_set_option(int(option), int(value)) != 0:
        raise RuntimeError(last_error())


def get_option(option: int) -> int:
    return int(lib.gsr_get_option(int(option)))


def set_stage_timing(enable: bool) -> None:
    lib.gsr_set_stage_timing(1 if enable else 0)


def call_times_ms(capacity: int = 256) -> list:
    """Device milliseconds (first kernel to last) of this thread's most recent timed calls, newest first."""
    arr = (ctypes.c_float * capacity)()
    n = lib.gsr_get_call_times(arr, capacity)
    if n < 0:
        raise RuntimeError(last_error())
    return [float(arr[i]) for i in range(n)]


def backward_times_ms() -> dict:
    """Mean milliseconds of gsr_backward's two kernels over the calls since ``set_stage_timing(True)`` (process-wide)."""
    arr = (ctypes.c_float * 2)()
    n = lib.gsr_get_backward_times(ctypes.byref(arr))
    if n < 0:
        raise RuntimeError(last_error())
    return {"render_backward": float(arr[0]), "preprocess_backward": float(arr[1]), "calls": int(n)}


def stage_times_ms() -> dict:
    """Mean per-stage milliseconds over the calls since ``set_stage_timing(True)``; key ``calls``
    holds how many calls were averaged."""
    arr = (ctypes.c_float * len(STAGES))()
    n = lib.gsr_get_stage_times(ctypes.byref(arr))
    if n < 0:
        raise RuntimeError(last_error())
    out = dict(zip(STAGES, (float(v) for v in arr)))
    out["calls"] = int(n)
    return out
