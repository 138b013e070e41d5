"""ctypes front-end of oracle/_ref/libgsr_ref_hip.so: the REFERENCE's own CUDA rasterizer compiled for gfx950
(oracle/build_ref_hip.py).  TEST / BENCH INFRASTRUCTURE ONLY: a measuring stick on the same GPU, imported by
tests/ and by bench.py's ``reference_hip`` leg; the product never imports it."""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libgsr_ref_hip.so")
_lib = None


def available() -> bool:
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        p = ctypes.c_void_p
        L.gsr_refhip_forward.restype = ctypes.c_int
        L.gsr_refhip_forward.argtypes = [ctypes.c_int] * 3 + [p, ctypes.c_int, ctypes.c_int, p, p, p, p, p, ctypes.c_float,
                                                             p, p, p, p, p, ctypes.c_float, ctypes.c_float, p, p, p, p]
        L.gsr_refhip_last_lists.restype = ctypes.c_int
        L.gsr_refhip_last_lists.argtypes = [ctypes.c_int] * 4 + [ctypes.POINTER(p)] * 3
        L.gsr_refhip_backward.restype = ctypes.c_int
        L.gsr_refhip_backward.argtypes = ([ctypes.c_int] * 4 + [p, ctypes.c_int, ctypes.c_int] + [p] * 4 + [ctypes.c_float]
                                          + [p] * 5 + [ctypes.c_float, ctypes.c_float] + [p] * 15)
        _lib = L
    return _lib


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def forward(cloud, cam, bg: torch.Tensor, outputs=None):
    """One forward call of the reference pipeline (default stream, like the reference).  ``cloud`` / ``cam`` / ``bg``
    on the GPU.  Returns (num_rendered, color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P])."""
    dev = cloud.means3D.device
    P, H, W = cloud.P, int(cam.image_height), int(cam.image_width)
    if outputs is None:
        outputs = (torch.zeros((3, H, W), device=dev), torch.zeros((1, H, W), device=dev), torch.zeros((1, H, W), device=dev),
                   torch.zeros((P,), dtype=torch.int32, device=dev))
    color, depth, alpha, radii = outputs
    M = 0 if cloud.shs is None else int(cloud.shs.shape[1])
    torch.cuda.current_stream(dev).synchronize()        # the reference runs on the legacy default stream
    n = lib().gsr_refhip_forward(P, int(cloud.sh_degree), M, bg.data_ptr(), W, H, cloud.means3D.data_ptr(), _ptr(cloud.shs),
                                 _ptr(cloud.colors_precomp), cloud.opacities.data_ptr(), _ptr(cloud.scales), 1.0,
                                 _ptr(cloud.rotations), None, cam.world_view_transform.data_ptr(),
                                 cam.full_proj_transform.data_ptr(), cam.camera_center.data_ptr(), float(cam.tanfovx),
                                 float(cam.tanfovy), color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), radii.data_ptr())
    torch.cuda.synchronize(dev)
    return n, color, depth, alpha, radii


def backward(cloud, cam, bg, n, radii, alpha, dL_dcolor, dL_ddepth, dL_dalpha):
    """Backward of the preceding ``forward`` call through the reference's own kernels; returns the gradient dict."""
    dev = cloud.means3D.device
    P, H, W = cloud.P, int(cam.image_height), int(cam.image_width)
    M = 0 if cloud.shs is None else int(cloud.shs.shape[1])
    z = lambda *s: torch.zeros(s, device=dev)
    g = {"means2D": z(P, 3), "conic": z(P, 4), "opacity": z(P, 1), "colors": z(P, 3), "depths": z(P, 1), "means3D": z(P, 3),
         "cov3D": z(P, 6), "sh": z(P, max(M, 1), 3), "scales": z(P, 3), "rotations": z(P, 4)}
    torch.cuda.synchronize(dev)
    rc = lib().gsr_refhip_backward(
        P, int(cloud.sh_degree), M, int(n), bg.data_ptr(), W, H, cloud.means3D.data_ptr(), _ptr(cloud.shs),
        _ptr(cloud.colors_precomp), _ptr(cloud.scales), 1.0, _ptr(cloud.rotations), None, cam.world_view_transform.data_ptr(),
        cam.full_proj_transform.data_ptr(), cam.camera_center.data_ptr(), float(cam.tanfovx), float(cam.tanfovy),
        radii.data_ptr(), alpha.data_ptr(), dL_dcolor.contiguous().data_ptr(), dL_ddepth.contiguous().data_ptr(),
        dL_dalpha.contiguous().data_ptr(), g["means2D"].data_ptr(), g["conic"].data_ptr(), g["opacity"].data_ptr(),
        g["colors"].data_ptr(), g["depths"].data_ptr(), g["means3D"].data_ptr(), g["cov3D"].data_ptr(), g["sh"].data_ptr(),
        g["scales"].data_ptr(), g["rotations"].data_ptr())
    if rc != 0:
        raise RuntimeError("gsr_refhip_backward: no preceding forward call")
    torch.cuda.synchronize(dev)
    return g
