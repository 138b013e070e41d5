"""ctypes front-end of oracle/_ref/libgsr_ref.so: the REFERENCE's own rasterizer sources compiled
for the host (see oracle/build_ref.py).  TEST INFRASTRUCTURE ONLY.

Same keyword interface and result dictionary as ``oracle.cpu_oracle.forward`` so the two can be
diffed key by key.  Available wherever the .so exists: it is built in the container that has
/root/reference mounted and travels to the GPU box as a git-ignored build artefact.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict

import numpy as np

from . import build_ref
from .cpu_oracle import _F, _I32, _U8, _U32, _U64, _f32, _ptr

_lib = None


def available() -> bool:
    return os.path.exists(build_ref.LIB) or os.path.isdir(build_ref.REFERENCE_DGR)


def lib():
    global _lib
    if _lib is None:
        if os.path.isdir(build_ref.REFERENCE_DGR):
            build_ref.build()
        L = ctypes.CDLL(build_ref.LIB)
        L.gsr_ref_forward.restype = ctypes.c_longlong
        L.gsr_ref_forward.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, _F, ctypes.c_int, ctypes.c_int,
            _F, _F, _F, _F, _F, ctypes.c_float, _F, _F, _F, _F, _F, ctypes.c_float, ctypes.c_float,
            _F, _F, _F, _I32, _F, _F, _F, _F, _U32, _U32, _U32, ctypes.c_size_t, _U64, _U32, _U32]
        L.gsr_ref_mark_visible.restype = None
        L.gsr_ref_mark_visible.argtypes = [ctypes.c_int, _F, _F, _F, _U8]
        _lib = L
    return _lib


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    m, v, p = _f32(means3D), _f32(viewmatrix), _f32(projmatrix)
    P = 0 if m is None else m.shape[0]
    out = np.zeros(P, dtype=np.uint8)
    if P:
        lib().gsr_ref_mark_visible(P, _ptr(m, _F), _ptr(v, _F), _ptr(p, _F), _ptr(out, _U8))
    return out.astype(bool)


def forward(*, means3D, opacities, bg, width: int, height: int, viewmatrix, projmatrix, campos, tanfovx: float,
            tanfovy: float, sh_degree: int = 0, scale_modifier: float = 1.0, shs=None, colors_precomp=None,
            scales=None, rotations=None, cov3D_precomp=None, intermediates: bool = False) -> Dict[str, np.ndarray]:
    m = _f32(means3D)
    P = 0 if m is None else int(m.shape[0])
    H, W = int(height), int(width)
    out = {"color": np.zeros((3, H, W), np.float32), "depth": np.zeros((1, H, W), np.float32),
           "alpha": np.zeros((1, H, W), np.float32), "radii": np.zeros(P, np.int32), "num_rendered": 0}
    if P == 0:
        return out
    sh, col, sc, rot, cov = _f32(shs), _f32(colors_precomp), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    M = 0 if sh is None else int(sh.shape[1])
    op, bgv, vm, pm, cp = _f32(opacities), _f32(bg), _f32(viewmatrix), _f32(projmatrix), _f32(campos)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    inter = {}
    if intermediates:
        inter = {"means2D": np.zeros((P, 2), np.float32), "depths": np.zeros(P, np.float32),
                 "conic_opacity": np.zeros((P, 4), np.float32), "rgb": np.zeros((P, 3), np.float32),
                 "tiles_touched": np.zeros(P, np.uint32), "point_offsets": np.zeros(P, np.uint32),
                 "n_contrib": np.zeros((H, W), np.uint32), "ranges": np.zeros((T, 2), np.uint32)}
    L = lib()

    def call(cap, keys, lst):
        return int(L.gsr_ref_forward(
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
        keys, lst = np.zeros(max(D, 1), np.uint64), np.zeros(max(D, 1), np.uint32)
        for k in ("color", "depth", "alpha"):
            out[k].fill(0)
        D = call(max(D, 1), keys, lst)
        inter["point_list_keys"], inter["point_list"] = keys[:D], lst[:D]
    out["num_rendered"] = D
    out.update(inter)
    return out


def backward(**kw) -> Dict[str, np.ndarray]:
    """Forward + backward through the reference's own sources (same interface as cpu_oracle.backward)."""
    from .cpu_oracle import _backward_argtypes, run_backward
    L = lib()
    if not getattr(L, "_bw_ready", False):
        L.gsr_ref_forward_backward.restype = ctypes.c_longlong
        L.gsr_ref_forward_backward.argtypes = _backward_argtypes()
        L._bw_ready = True
    return run_backward(L.gsr_ref_forward_backward, **kw)
