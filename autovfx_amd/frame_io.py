"""Frame outputs the reference's frame loop writes (``scene_representation.py:425-438``).

* ``images/<name>.png``  RGBA8, quantised like ``torchvision.utils.save_image``: ``clamp(x*255 + 0.5, 0, 255)``
  truncated (the fused ``pack_rgba8`` kernel) -- this is the file ``blend_all.py`` composites over;
* ``depth/<name>.npy``   float32 ``[H,W]`` un-normalised accumulated depth;
* ``normal/<name>.png``  ``uint8((n + 1) / 2 * 255)`` (truncation), RGB.

* ``depth/<name>.png``   the turbo-coloured preview of the depth map the reference's depth video is made of
  (``depth2img(depth, scale=3.0)``, ``sugar/gaussian_splatting/render.py:45-49``: ``uint8(clip(depth / 3, 0, 1) * 255)``
  through ``cv2.COLORMAP_TURBO``).

The PNG encoder is a dependency-free one (zlib + CRC from the standard library): ``torchvision``, ``cv2`` and
``imageio`` are not installed here.  Nor can the turbo table be read out of OpenCV here: ``TURBO_LUT`` is the published
256-entry turbo colour map (the float table OpenCV's ``colormap.cpp`` and matplotlib both carry) times 255, rounded to
nearest as ``convertTo(CV_8U, 255)`` does -- no entry lies within 1e-3 of a rounding boundary; ``scripts/make_turbo_lut.py``
regenerates it and ``tests/test_frame_io.py`` compares it with matplotlib's copy of the table where that is importable.
"""
from __future__ import annotations

import os
import struct
import zlib

import numpy as np
import torch

from .frame_parallel import pack_rgba8

# uint8 [256, 3] RGB, see the module docstring
TURBO_LUT = np.frombuffer(__import__("base64").b64decode(
    "MBI7MhVDMxhKNBtRNR5YNiFfNyRmOCdtOSpzOi15Oy+APDKGPTWLPjiRPzuXPz6cQECiQUOnQUasQkmxQku1Q066RFG/RFTDRFbHRVnLRVzPRV7TRmHWRmTa"
    "RmbdRmngRmvjR27mR3HpR3PrR3buR3jwR3vyRn30RoD2RoL4RoX6Rof7RYr8RYz9RI/+Q5H+QpT/QZb/QJn/Ppv+PZ7+O6D9OqP8OKX7N6j6Nav4M633Ma/1"
    "L7L0LrTyLLfwKrnuKLzrJ77pJcDnI8PkIsXiIMffH8ndHsvaHM3YG9DVGtLSGtTQGdXNGNfKGNnIGNvFGN3CGN7AGOC9GeK7GeO5GuS2HOa0HeeyH+mvIOqs"
    "IuuqJeynJ+6kKu+hLPCeL/GbMvKYNfOUOPSRPPWOP/aKQ/eHRviESviATvl9Uvp6Vfp2WftzXfxvYfxsZf1paf1mbf5icf5fdf5cef5Zff9WgP9ThP9RiP9O"
    "i/9Lj/9Jkv9Hlv5Emf5CnP5An/0/of09pPw8p/w6qfs5rPs4r/o3sfk2tPg2t/c1ufY1vPU0vvQ0wfM0w/E0xvA0yO80y+00zew00Oo00uk11Oc11+U12eQ2"
    "2+I23eA339834d0349s45dk459c56dU569M57NE67s8678068cs68sk69Mc69cU69sM698E6+L45+bw5+ro5+7g4+7Y3/LM2/LE2/a41/aw0/qkz/qcy/qQx"
    "/qEw/p4v/pst/pks/pYr/pMq/pAp/Y0n/Yom/Icl/IQj+4Ei+34h+nsf+Xge+XUd+HIc928a9mwZ9WkY9GYX82MV8mAU8V0T8FsS71gR7VUQ7FMP61AO6k4N"
    "6EsM50kM5UcL5EUK4kMK4UEJ3z8I3T0I3DsH2jkH2DcG1jUG1DMF0jEF0C8Fzi0EzCsEyioEyCgDxSYDwyUDwSMCviECvCACuR4Ctx0CtBsBshoBrxgBrBcB"
    "qRYBpxQBpBMBoRIBnhABmw8BmA4BlQ0BkgsBjgoBiwkCiAgChQcCgQYCfgUCegQD"), np.uint8).reshape(256, 3)


def depth2img(depth: np.ndarray, scale: float = 3.0) -> np.ndarray:
    """``depth2img`` of the reference (``sugar/gaussian_splatting/render.py:45-49``, called with ``scale=3.0`` at
    ``scene_representation.py:432``) -> uint8 ``[H,W,3]`` RGB, the colours ``cv2.imwrite`` puts into the PNG."""
    d = np.clip(np.asarray(depth) / scale, 0.0, 1.0)
    return TURBO_LUT[(d * 255).astype(np.uint8)]


def encode_png(image: np.ndarray, compress_level: int = 3) -> bytes:
    """uint8 ``[H,W,3]`` or ``[H,W,4]`` -> PNG bytes (8-bit truecolour, no interlace, filter 0)."""
    if image.dtype != np.uint8 or image.ndim != 3 or image.shape[2] not in (3, 4):
        raise ValueError("encode_png expects uint8 [H,W,3|4]")
    h, w, c = image.shape
    raw = np.concatenate((np.zeros((h, 1), np.uint8), np.ascontiguousarray(image).reshape(h, w * c)), axis=1).tobytes()

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    ihdr = struct.pack(">IIBBBBB", w, h, 8, 6 if c == 4 else 2, 0, 0, 0)
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw, compress_level))
            + chunk(b"IEND", b""))


def decode_png(data: bytes) -> np.ndarray:
    """Inverse of ``encode_png`` for the PNGs it writes (filter type 0 only)."""
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", 0
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            c = {2: 3, 6: 4}[ctype]
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * c)
    assert not rows[:, 0].any(), "only filter type 0 is supported"
    return rows[:, 1:].reshape(h, w, c).copy()


def _frame_paths(out_dir: str, name: str) -> dict:
    paths = {k: os.path.join(out_dir, k, name + ext) for k, ext in (("images", ".png"), ("depth", ".npy"), ("normal", ".png"))}
    paths["depth_preview"] = os.path.join(out_dir, "depth", name + ".png")
    for p in paths.values():
        os.makedirs(os.path.dirname(p), exist_ok=True)
    return paths


def _frame_to_host(result: dict):
    """What the files hold, as host arrays: the quantisation runs where the tensors live (on the GPU: the fused
    ``pack_rgba8`` kernel), only bytes that end up in the files cross to the host."""
    rgba = result["render"]
    rgba8 = pack_rgba8(rgba[:3], rgba[3:4]).permute(1, 2, 0).contiguous().cpu().numpy()
    depth = result["depth"].detach().to(torch.float32).cpu().numpy()
    normal = ((result["normal"].detach() + 1.0) / 2.0 * 255.0).to(torch.uint8).cpu().numpy()   # truncation, as the reference
    return rgba8, depth, normal


def _write_host_frame(paths: dict, rgba8: np.ndarray, depth: np.ndarray, normal: np.ndarray, compress_level: int = 3) -> dict:
    with open(paths["images"], "wb") as f:
        f.write(encode_png(rgba8, compress_level))
    np.save(paths["depth"], depth)
    with open(paths["depth_preview"], "wb") as f:   # (scene_representation.py:432-433)
        f.write(encode_png(depth2img(depth.squeeze(), 3.0), compress_level))
    with open(paths["normal"], "wb") as f:
        f.write(encode_png(normal, compress_level))
    return paths


def write_frame_outputs(out_dir: str, name: str, result: dict) -> dict:
    """Write the four files of one frame from a ``render()`` result dict; returns their paths."""
    return _write_host_frame(_frame_paths(out_dir, name), *_frame_to_host(result))


class FrameWriter:
    """The same four files per frame, encoded and written by a pool of host threads behind the rendering loop.

    ``submit`` quantises on the device and copies the frame to the host (in the caller's thread and stream: the frame is
    complete when it returns, the tensors may be reused), then hands the host arrays to a worker; zlib and file writes
    release the interpreter lock, so the workers run beside each other and beside the loop.  The reference writes its
    frames inline (``scene_representation.py:425-438``: ``save_image`` + ``np.save`` per frame), which at 960x540 costs
    more host time per frame than the GPU needs for two hundred frames.  ``close()`` (or leaving the ``with`` block)
    waits for everything and re-raises the first error."""

    def __init__(self, out_dir: str, workers: int = 0, compress_level: int = 3, max_pending: int = 0):
        from concurrent.futures import ThreadPoolExecutor
        if workers <= 0:
            workers = max(1, min(32, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
        self.out_dir, self.compress_level = out_dir, compress_level
        self._pool = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="frame-writer")
        self._pending = []
        self._max_pending = max_pending if max_pending > 0 else 4 * workers   # bounds the host memory held by queued frames

    def submit(self, name: str, result: dict) -> None:
        arrays = _frame_to_host(result)
        paths = _frame_paths(self.out_dir, name)
        while len(self._pending) >= self._max_pending:
            self._pending.pop(0).result()
        self._pending.append(self._pool.submit(_write_host_frame, paths, *arrays, self.compress_level))

    def close(self) -> None:
        pending, self._pending = self._pending, []
        try:
            for f in pending:
                f.result()
        finally:
            self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
