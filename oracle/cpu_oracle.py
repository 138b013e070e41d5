"""ctypes front-end of the CPU oracle (oracle/gsr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of bench.py.  The product (autovfx_amd/, diff_gaussian_rasterization/) never
imports it.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgsr_oracle.so")
_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_U8 = ctypes.POINTER(ctypes.c_uint8)
_U32 = ctypes.POINTER(ctypes.c_uint32)
_U64 = ctypes.POINTER(ctypes.c_uint64)
_I32 = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gsr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgsr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.gsro_key_bits.restype = ctypes.c_uint32
        L.gsro_key_bits.argtypes = [ctypes.c_uint32]
        L.gsro_mark_visible.restype = None
        L.gsro_mark_visible.argtypes = [ctypes.c_int, _F, _F, _F, _U8]
        L.gsro_forward.restype = ctypes.c_int64
        L.gsro_forward.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int,  # P deg M bg W H
            _F, _F, _F, _F, _F, ctypes.c_float, _F, _F,                                # means3D shs colors opac scales mod rots cov3D
            _F, _F, _F, ctypes.c_float, ctypes.c_float,                                # view proj cam tanx tany
            _F, _F, _F, _I32,                                                          # out_color out_depth out_alpha radii
            _F, _F, _F, _F, _U32, _U32, _U32,                                          # means2D depths conop rgb tiles offsets ncontrib
            ctypes.c_size_t, _U64, _U32, _U32]                                         # cap keys list ranges
        _lib = L
    return _lib


def _f32(a) -> Optional[np.ndarray]:
    if a is None:
        return None
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if a.size else None


def _ptr(a, ty):
    return ctypes.cast(None, ty) if a is None else a.ctypes.data_as(ty)


def key_bits(num_tiles: int) -> int:
    return int(lib().gsro_key_bits(num_tiles))


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    m, v, p = _f32(means3D), _f32(viewmatrix), _f32(projmatrix)
    P = 0 if m is None else m.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    if P:
        lib().gsro_mark_visible(P, _ptr(m, _F), _ptr(v, _F), _ptr(p, _F), _ptr(out, _U8))
    return out.astype(bool)


def forward(*, means3D, opacities, bg, width: int, height: int, viewmatrix, projmatrix, campos,
            tanfovx: float, tanfovy: float, sh_degree: int = 0, scale_modifier: float = 1.0, shs=None,
            colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            intermediates: bool = False) -> Dict[str, np.ndarray]:
    """Run the restated forward pass; returns color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P],
    num_rendered and (optionally) every intermediate of the pipeline."""
    m = _f32(means3D)
    P = 0 if m is None else int(m.shape[0])
    H, W = int(height), int(width)
    out = {
        "color": np.zeros((3, H, W), np.float32), "depth": np.zeros((1, H, W), np.float32),
        "alpha": np.zeros((1, H, W), np.float32), "radii": np.zeros(P, np.int32), "num_rendered": 0,
    }
    if P == 0:
        return out
    sh, col, sc, rot, cov = _f32(shs), _f32(colors_precomp), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    if (sh is None) == (col is None):
        raise ValueError("exactly one of shs / colors_precomp")
    if ((sc is None or rot is None) and cov is None) or ((sc is not None or rot is not None) and cov is not None):
        raise ValueError("exactly one of (scales, rotations) / cov3D_precomp")
    M = 0 if sh is None else int(sh.shape[1])
    op, bgv = _f32(opacities), _f32(bg)
    vm, pm, cp = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    inter = {}
    cap = 0
    keys = lst = None
    if intermediates:
        inter = {"means2D": np.zeros((P, 2), np.float32), "depths": np.zeros(P, np.float32),
                 "conic_opacity": np.zeros((P, 4), np.float32), "rgb": np.zeros((P, 3), np.float32),
                 "tiles_touched": np.zeros(P, np.uint32), "point_offsets": np.zeros(P, np.uint32),
                 "n_contrib": np.zeros((H, W), np.uint32), "ranges": np.zeros((T, 2), np.uint32)}
    L = lib()

    def call(cap, keys, lst):
        return int(L.gsro_forward(
            P, int(sh_degree), M, _ptr(bgv, _F), W, H, _ptr(m, _F), _ptr(sh, _F), _ptr(col, _F), _ptr(op, _F),
            _ptr(sc, _F), float(scale_modifier), _ptr(rot, _F), _ptr(cov, _F), _ptr(vm, _F), _ptr(pm, _F),
            _ptr(cp, _F), float(tanfovx), float(tanfovy), _ptr(out["color"], _F), _ptr(out["depth"], _F),
            _ptr(out["alpha"], _F), _ptr(out["radii"], _I32), _ptr(inter.get("means2D"), _F),
            _ptr(inter.get("depths"), _F), _ptr(inter.get("conic_opacity"), _F), _ptr(inter.get("rgb"), _F),
            _ptr(inter.get("tiles_touched"), _U32), _ptr(inter.get("point_offsets"), _U32),
            _ptr(inter.get("n_contrib"), _U32), cap, _ptr(keys, _U64), _ptr(lst, _U32),
            _ptr(inter.get("ranges"), _U32)))

    D = call(0, None, None)
    if intermediates:
        keys = np.zeros(max(D, 1), np.uint64)
        lst = np.zeros(max(D, 1), np.uint32)
        for k in ("color", "depth", "alpha"):
            out[k].fill(0)
        D = call(max(D, 1), keys, lst)
        inter["point_list_keys"] = keys[:D]
        inter["point_list"] = lst[:D]
    out["num_rendered"] = D
    out.update(inter)
    return out


_BACKWARD_ARGTYPES = None


def _backward_argtypes():
    return [ctypes.c_int, ctypes.c_int, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int,   # P deg M bg W H
            _F, _F, _F, _F, _F, ctypes.c_float, _F, _F,                                  # means3D shs colors opac scales mod rots cov3D
            _F, _F, _F, ctypes.c_float, ctypes.c_float,                                  # view proj cam tanx tany
            _F, _F, _F,                                                                  # dL_dout color depth alpha
            _F, _F, _F, _I32,                                                            # out_color out_depth out_alpha radii
            _F, _F, _F, _F, _F, _F, _F, _F, _F, _F]                                      # grads


def run_backward(fn, *, means3D, opacities, bg, width, height, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                 dL_dcolor, dL_ddepth, dL_dalpha, sh_degree=0, scale_modifier=1.0, shs=None, colors_precomp=None,
                 scales=None, rotations=None, cov3D_precomp=None) -> Dict[str, np.ndarray]:
    """Forward + backward through a ``*_forward_backward`` C entry point (this oracle's or oracle/_ref's).
    Returns the forward outputs and every gradient the reference's backward produces."""
    m = _f32(means3D)
    P = 0 if m is None else int(m.shape[0])
    H, W = int(height), int(width)
    sh, col, sc, rot, cov = _f32(shs), _f32(colors_precomp), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    M = 0 if sh is None else int(sh.shape[1])
    out = {"color": np.zeros((3, H, W), np.float32), "depth": np.zeros((1, H, W), np.float32),
           "alpha": np.zeros((1, H, W), np.float32), "radii": np.zeros(P, np.int32),
           "dL_dmeans2D": np.zeros((P, 3), np.float32), "dL_dcolors": np.zeros((P, 3), np.float32),
           "dL_dopacity": np.zeros((P, 1), np.float32), "dL_dmeans3D": np.zeros((P, 3), np.float32),
           "dL_dcov3D": np.zeros((P, 6), np.float32), "dL_dsh": np.zeros((P, M, 3), np.float32),
           "dL_dscales": np.zeros((P, 3), np.float32), "dL_drotations": np.zeros((P, 4), np.float32),
           "dL_dconic": np.zeros((P, 4), np.float32), "dL_ddepths": np.zeros((P, 1), np.float32)}
    if P == 0:
        out["num_rendered"] = 0
        return out
    op, bgv, vm, pm, cp = _f32(opacities), _f32(bg), _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    gc, gd, ga = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
    n = fn(P, int(sh_degree), M, _ptr(bgv, _F), W, H, _ptr(m, _F), _ptr(sh, _F), _ptr(col, _F), _ptr(op, _F),
           _ptr(sc, _F), float(scale_modifier), _ptr(rot, _F), _ptr(cov, _F), _ptr(vm, _F), _ptr(pm, _F), _ptr(cp, _F),
           float(tanfovx), float(tanfovy), _ptr(gc, _F), _ptr(gd, _F), _ptr(ga, _F), _ptr(out["color"], _F),
           _ptr(out["depth"], _F), _ptr(out["alpha"], _F), _ptr(out["radii"], _I32), _ptr(out["dL_dmeans2D"], _F),
           _ptr(out["dL_dcolors"], _F), _ptr(out["dL_dopacity"], _F), _ptr(out["dL_dmeans3D"], _F),
           _ptr(out["dL_dcov3D"], _F), _ptr(out["dL_dsh"], _F), _ptr(out["dL_dscales"], _F),
           _ptr(out["dL_drotations"], _F), _ptr(out["dL_dconic"], _F), _ptr(out["dL_ddepths"], _F))
    out["num_rendered"] = int(n)
    return out


def backward(**kw) -> Dict[str, np.ndarray]:
    L = lib()
    if not getattr(L, "_bw_ready", False):
        L.gsro_forward_backward.restype = ctypes.c_int64
        L.gsro_forward_backward.argtypes = _backward_argtypes()
        L._bw_ready = True
    return run_backward(L.gsro_forward_backward, **kw)


_D = ctypes.POINTER(ctypes.c_double)


def backward_f64(*, means3D, opacities, bg, width, height, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                 dL_dcolor, dL_ddepth, dL_dalpha, sh_degree=0, scale_modifier=1.0, shs=None, colors_precomp=None,
                 scales=None, rotations=None, cov3D_precomp=None) -> Dict[str, np.ndarray]:
    """The gradient TRUTH (gsr_oracle.c: gsro_backward_f64): the reference's backward formulas evaluated in double on
    the fp32 forward state.  Same keys as ``backward`` for the gradients, all float64; plus ``abs_sums`` [P,10] (the sums of
    |term| behind the ten per-Gaussian sums of the per-pixel pass, see ``fp32_noise``) and ``radii``."""
    L = lib()
    if not getattr(L, "_bw64_ready", False):
        L.gsro_backward_f64.restype = ctypes.c_int64
        L.gsro_backward_f64.argtypes = _backward_argtypes()[:22] + [_D] * 11 + [_I32]
        L._bw64_ready = True
    m = _f32(means3D)
    P = 0 if m is None else int(m.shape[0])
    H, W = int(height), int(width)
    sh, col, sc, rot, cov = _f32(shs), _f32(colors_precomp), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    M = 0 if sh is None else int(sh.shape[1])
    z = lambda *s: np.zeros(s, np.float64)
    out = {"dL_dmeans2D": z(P, 3), "dL_dcolors": z(P, 3), "dL_dopacity": z(P, 1), "dL_dmeans3D": z(P, 3), "dL_dcov3D": z(P, 6),
           "dL_dsh": z(P, M, 3), "dL_dscales": z(P, 3), "dL_drotations": z(P, 4), "dL_dconic": z(P, 4), "dL_ddepths": z(P, 1),
           "abs_sums": z(P, 10), "radii": np.zeros(P, np.int32)}
    if P == 0:
        return out
    op, bgv, vm, pm, cp = _f32(opacities), _f32(bg), _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    gc, gd, ga = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
    if gc is None:
        gc = np.zeros((3, H, W), np.float32)
    if gd is None:
        gd = np.zeros((1, H, W), np.float32)
    if ga is None:
        ga = np.zeros((1, H, W), np.float32)
    L.gsro_backward_f64(P, int(sh_degree), M, _ptr(bgv, _F), W, H, _ptr(m, _F), _ptr(sh, _F), _ptr(col, _F), _ptr(op, _F),
                        _ptr(sc, _F), float(scale_modifier), _ptr(rot, _F), _ptr(cov, _F), _ptr(vm, _F), _ptr(pm, _F),
                        _ptr(cp, _F), float(tanfovx), float(tanfovy), _ptr(gc, _F), _ptr(gd, _F), _ptr(ga, _F),
                        _ptr(out["dL_dmeans2D"], _D), _ptr(out["dL_dcolors"], _D), _ptr(out["dL_dopacity"], _D),
                        _ptr(out["dL_dmeans3D"], _D), _ptr(out["dL_dcov3D"], _D), _ptr(out["dL_dsh"], _D),
                        _ptr(out["dL_dscales"], _D), _ptr(out["dL_drotations"], _D), _ptr(out["dL_dconic"], _D),
                        _ptr(out["dL_ddepths"], _D), _ptr(out["abs_sums"], _D), _ptr(out["radii"], _I32))
    return out


def fp32_noise(truth: Dict[str, np.ndarray], *, means3D, width, height, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
               sh_degree=0, scale_modifier=1.0, shs=None, scales=None, rotations=None, cov3D_precomp=None, samples: int = 8,
               ulps: float = 2.0, seed: int = 0, **_unused) -> Dict[str, np.ndarray]:
    """How far may an fp32 evaluation of a gradient be from the truth?  A conditioning-aware yardstick, per element.

    Every fp32 backward forms the ten per-Gaussian sums of the per-pixel pass with rounding errors: each term carries a few
    relative roundings and the additions round again, so a sum S = sum t_i comes out as S + e with |e| of the order of
    2^-24 * sum |t_i| -- NOT 2^-24 * |S| when the terms cancel.  The per-Gaussian chain (backward.cu:144-413) then maps the sums
    to the parameter gradients, and for a needle or a splat at the near plane it amplifies e by 1e3 .. 1e8.  This function
    draws ``samples`` such error vectors (e = +-ulps * 2^-24 * sum |t_i| with random signs, fixed seed; ``abs_sums`` from
    ``backward_f64``), rounds the perturbed sums to fp32, runs the REFERENCE's fp32 chain on them (gsro_preprocess_backward,
    backward.cu's arithmetic op for op) and returns, per gradient array, the largest |result - truth| seen per element.  It is a
    pure function of the scene: the same yardstick on every box."""
    L = lib()
    if not getattr(L, "_pb_ready", False):
        L.gsro_preprocess_backward.restype = None
        L.gsro_preprocess_backward.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _F, _I32, _F, _F, _F, ctypes.c_float, _F, _F, _F, _F,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, _F, _F, _F, _F, _F, _F, _F, _F, _F]
        L._pb_ready = True
    m = _f32(means3D)
    P = 0 if m is None else int(m.shape[0])
    sh, sc, rot, cov = _f32(shs), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    M = 0 if sh is None else int(sh.shape[1])
    vm, pm, cp = _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    keys = ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    worst = {k: np.zeros(truth[k].shape, np.float64) for k in keys}
    if P == 0:
        return worst
    A = truth["abs_sums"]
    rng = np.random.default_rng(seed)
    eps = ulps * 2.0 ** -24
    radii = np.ascontiguousarray(truth["radii"], np.int32)
    for _ in range(samples):
        sign = rng.integers(0, 2, size=(P, 10)).astype(np.float64) * 2.0 - 1.0
        e = sign * eps * A
        g2 = np.zeros((P, 3), np.float32); g2[:, :2] = truth["dL_dmeans2D"][:, :2] + e[:, 0:2]
        gcn = np.zeros((P, 4), np.float32); gcn[:, (0, 1, 3)] = truth["dL_dconic"][:, (0, 1, 3)] + e[:, 2:5]
        gop = np.ascontiguousarray(truth["dL_dopacity"].reshape(P) + e[:, 5], np.float32)
        gcol = np.ascontiguousarray(truth["dL_dcolors"] + e[:, 6:9], np.float32)
        gdep = np.ascontiguousarray(truth["dL_ddepths"].reshape(P) + e[:, 9], np.float32)
        o = {"dL_dmeans3D": np.zeros((P, 3), np.float32), "dL_dcov3D": np.zeros((P, 6), np.float32),
             "dL_dsh": np.zeros((P, M, 3), np.float32), "dL_dscales": np.zeros((P, 3), np.float32), "dL_drotations": np.zeros((P, 4), np.float32)}
        L.gsro_preprocess_backward(P, int(sh_degree), M, _ptr(m, _F), _ptr(radii, _I32), _ptr(sh, _F), _ptr(sc, _F), _ptr(rot, _F),
                                   float(scale_modifier), _ptr(cov, _F), _ptr(vm, _F), _ptr(pm, _F), _ptr(cp, _F), int(width), int(height),
                                   float(tanfovx), float(tanfovy), _ptr(g2, _F), _ptr(gcn, _F), _ptr(gcol, _F), _ptr(gdep, _F),
                                   _ptr(o["dL_dmeans3D"], _F), _ptr(o["dL_dcov3D"], _F), _ptr(o["dL_dsh"] if M else None, _F),
                                   _ptr(o["dL_dscales"], _F), _ptr(o["dL_drotations"], _F))
        o["dL_dmeans2D"], o["dL_dopacity"], o["dL_dcolors"] = g2, gop.reshape(P, 1), gcol
        for k in keys:
            with np.errstate(invalid="ignore", over="ignore"):
                d = np.abs(o[k].astype(np.float64).reshape(truth[k].shape) - truth[k])
            worst[k] = np.fmax(worst[k], np.nan_to_num(d, nan=0.0, posinf=0.0))
    return worst
