"""Frame outputs the reference's frame loop writes (``scene_representation.py:425-438``).

* ``images/<name>.png``  RGBA8, quantised like ``torchvision.utils.save_image``: ``clamp(x*255 + 0.5, 0, 255)``
  truncated (the fused ``pack_rgba8`` kernel) -- this is the file ``blend_all.py`` composites over;
* ``depth/<name>.npy``   float32 ``[H,W]`` un-normalised accumulated depth;
* ``normal/<name>.png``  ``uint8((n + 1) / 2 * 255)`` (truncation), RGB.

* ``depth/<name>.png``   the turbo-coloured preview of the depth map the reference's depth video is made of
  (``depth2img(depth, scale=3.0)``, ``sugar/gaussian_splatting/render.py:45-49``: ``uint8(clip(depth / 3, 0, 1) * 255)``
  through ``cv2.COLORMAP_TURBO``).

Two ways to the files.  ``FrameWriter`` / ``write_frame_outputs``: the frame crosses to the host as pixels and a pool of host
threads deflates it (zlib level 3) -- small files, 6.8 ms of host time per 960x540 frame even on 32 threads.  ``GpuFrameWriter``:
the FILE IMAGES are made on the GPU (``gsr_png_encode``: stored deflate blocks, Adler-32 and CRC-32 in the kernel; the .npy is a
constant header in front of the fp32 plane), one device-to-host copy per frame into pinned memory, and the host threads only
``write()`` -- files as large as the raw frames (7.3 MB per 960x540 frame instead of ~2), every reader sees the same pixels.

The host PNG encoder is a dependency-free one (zlib + CRC from the standard library): ``torchvision``, ``cv2`` and
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
    """A small PNG reader for 8-bit truecolour (+ alpha), non-interlaced files: what ``encode_png`` and the GPU encoders write (filter
    type 0 everywhere, or Paeth-filtered rows), any of the five filter types in general."""
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
    if not rows[:, 0].any():
        return rows[:, 1:].reshape(h, w, c).copy()
    out = np.zeros((h, w, c), np.int32)
    zero_row = np.zeros((w, c), np.int32)
    for y in range(h):
        ft, line = int(rows[y, 0]), rows[y, 1:].reshape(w, c).astype(np.int32)
        up = out[y - 1] if y else zero_row
        if ft == 0:
            out[y] = line
        elif ft == 1:
            out[y] = np.cumsum(line, axis=0) & 255
        elif ft == 2:
            out[y] = (line + up) & 255
        elif ft in (3, 4):
            left, ul = np.zeros(c, np.int32), np.zeros(c, np.int32)
            for x in range(w):
                b = up[x]
                if ft == 3:
                    pred = (left + b) >> 1
                else:
                    p = left + b - ul
                    pa, pb, pc = np.abs(p - left), np.abs(p - b), np.abs(p - ul)
                    pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, b, ul))
                left = (line[x] + pred) & 255
                out[y, x] = left
                ul = b
        else:
            raise ValueError(f"PNG filter type {ft}")
    return out.astype(np.uint8)


def _frame_paths(out_dir: str, name: str, made: "set | None" = None) -> dict:
    """The four paths of a frame; their directories are created.  ``made``: a writer's own memory of the directories it has
    created (once per directory and writer, not four system calls per frame).  It is per writer, not per process: a directory
    that is removed and re-rendered into later in the same process is created again by the next writer."""
    paths = {k: os.path.join(out_dir, k, name + ext) for k, ext in (("images", ".png"), ("depth", ".npy"), ("normal", ".png"))}
    paths["depth_preview"] = os.path.join(out_dir, "depth", name + ".png")
    for p in paths.values():
        d = os.path.dirname(p)
        if made is None or d not in made:
            os.makedirs(d, exist_ok=True)
            if made is not None:
                made.add(d)
    return paths


def _open_for_write(path: str):
    """``open(path, 'wb')``; a directory that vanished since the writer created it (results cleaned while a loop runs) is made again."""
    try:
        return open(path, "wb")
    except FileNotFoundError:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        return open(path, "wb")


def _frame_to_host(result: dict):
    """What the files hold, as host arrays: the quantisation runs where the tensors live (on the GPU: the fused
    ``pack_rgba8`` kernel), only bytes that end up in the files cross to the host."""
    rgba = result["render"]
    rgba8 = pack_rgba8(rgba[:3], rgba[3:4]).permute(1, 2, 0).contiguous().cpu().numpy()
    depth = result["depth"].detach().to(torch.float32).cpu().numpy()
    normal = ((result["normal"].detach() + 1.0) / 2.0 * 255.0).to(torch.uint8).cpu().numpy()   # truncation, as the reference
    return rgba8, depth, normal


def _write_host_frame(paths: dict, rgba8: np.ndarray, depth: np.ndarray, normal: np.ndarray, compress_level: int = 3) -> dict:
    with _open_for_write(paths["images"]) as f:
        f.write(encode_png(rgba8, compress_level))
    with _open_for_write(paths["depth"]) as f:
        np.save(f, depth)
    with _open_for_write(paths["depth_preview"]) as f:   # (scene_representation.py:432-433)
        f.write(encode_png(depth2img(depth.squeeze(), 3.0), compress_level))
    with _open_for_write(paths["normal"]) as f:
        f.write(encode_png(normal, compress_level))
    return paths


def write_frame_outputs(out_dir: str, name: str, result: dict) -> dict:
    """Write the four files of one frame from a ``render()`` result dict; returns their paths."""
    return _write_host_frame(_frame_paths(out_dir, name), *_frame_to_host(result))


def npy_header(shape, dtype=np.float32) -> bytes:
    """The bytes ``np.save`` puts in front of a C-ordered array of that shape and dtype (format 1.0, padded to a multiple of 64)."""
    import io
    f = io.BytesIO()
    np.lib.format.write_array_header_1_0(f, {"descr": np.lib.format.dtype_to_descr(np.dtype(dtype)), "fortran_order": False,
                                            "shape": tuple(int(v) for v in shape)})
    return f.getvalue()


def deflate_default() -> bool:
    """Compressed PNGs unless ``GSR_PNG_DEFLATE=0`` (then: stored deflate blocks, files as large as the raw image)."""
    return os.environ.get("GSR_PNG_DEFLATE", "1") not in ("0", "", "false", "False", "off")


def png_mode() -> str:
    return "deflate" if deflate_default() else "stored"


def png_size(width: int, height: int, channels: int) -> int:
    from . import _lib
    n = int(_lib.lib.gsr_png_size(int(width), int(height), int(channels)))
    if n == 0:
        raise ValueError(f"a {width}x{height} image with {channels} channels cannot be encoded")
    return n


def png_room(width: int, height: int, channels: int) -> int:
    """Bytes the encoder's output buffer must hold: the file, then the kernels' partial checksums (``gsr_png_room``)."""
    from . import _lib
    return int(_lib.lib.gsr_png_room(int(width), int(height), int(channels)))


def png_deflate_max_size(width: int, height: int, channels: int) -> int:
    from . import _lib
    n = int(_lib.lib.gsr_png_deflate_max_size(int(width), int(height), int(channels)))
    if n == 0:
        raise ValueError(f"a {width}x{height} image with {channels} channels cannot be encoded")
    return n


def png_deflate_room(width: int, height: int, channels: int) -> int:
    from . import _lib
    return int(_lib.lib.gsr_png_deflate_room(int(width), int(height), int(channels)))


def png_deflate_scratch(width: int, height: int, channels: int) -> int:
    from . import _lib
    return int(_lib.lib.gsr_png_deflate_scratch(int(width), int(height), int(channels)))


def encode_png_gpu_deflate_queued(image: torch.Tensor, planar: bool = False):
    """``encode_png_gpu_deflate`` without the host synchronisation: ``(out, length)`` -- the buffer the file is being written into
    (``png_deflate_room`` bytes) and a device ``int64[1]`` that will hold the file's length -- for callers that copy both out behind
    the kernels and look at them once their stream has drained."""
    import ctypes
    from . import _lib
    if not (image.is_cuda and image.dtype == torch.uint8 and image.dim() == 3):
        raise ValueError("encode_png_gpu_deflate expects a uint8 GPU tensor [H,W,C] or [C,H,W]")
    img = image.contiguous()
    C, H, W = (int(v) for v in (img.shape if planar else (img.shape[2], img.shape[0], img.shape[1])))
    room = png_deflate_room(W, H, C)
    if room == 0:
        raise ValueError(f"a {W}x{H} image with {C} channels cannot be encoded")
    out = torch.empty(room, dtype=torch.uint8, device=img.device)
    scratch = torch.empty(png_deflate_scratch(W, H, C), dtype=torch.uint8, device=img.device)
    length = torch.zeros(1, dtype=torch.int64, device=img.device)
    with torch.cuda.device(img.device):
        rc = _lib.lib.gsr_png_encode_deflate(img.data_ptr(), W, H, C, 1 if planar else 0, out.data_ptr(), scratch.data_ptr(), length.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream(img.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gsr_png_encode_deflate failed ({rc}): {_lib.last_error()}")
    return out[:png_deflate_max_size(W, H, C)], length


def encode_png_gpu_deflate(image: torch.Tensor, planar: bool = False) -> torch.Tensor:
    """``encode_png_gpu`` with a compressed IDAT (``gsr_png_encode_deflate``: Paeth filter, run-length matches, one Huffman code per
    image built on the GPU).  The file's length depends on the image, so this convenience form reads it back (one host
    synchronisation); the frame writer keeps it on the device and copies it out with the file."""
    import ctypes
    from . import _lib
    if not (image.is_cuda and image.dtype == torch.uint8 and image.dim() == 3):
        raise ValueError("encode_png_gpu_deflate expects a uint8 GPU tensor [H,W,C] or [C,H,W]")
    img = image.contiguous()
    C, H, W = (int(v) for v in (img.shape if planar else (img.shape[2], img.shape[0], img.shape[1])))
    room = png_deflate_room(W, H, C)
    if room == 0:
        raise ValueError(f"a {W}x{H} image with {C} channels cannot be encoded")
    out = torch.empty(room, dtype=torch.uint8, device=img.device)
    scratch = torch.empty(png_deflate_scratch(W, H, C), dtype=torch.uint8, device=img.device)
    length = torch.zeros(1, dtype=torch.int64, device=img.device)
    with torch.cuda.device(img.device):
        rc = _lib.lib.gsr_png_encode_deflate(img.data_ptr(), W, H, C, 1 if planar else 0, out.data_ptr(), scratch.data_ptr(), length.data_ptr(),
                                             ctypes.c_void_p(torch.cuda.current_stream(img.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gsr_png_encode_deflate failed ({rc}): {_lib.last_error()}")
    n = int(length.item())
    assert 0 < n <= png_deflate_max_size(W, H, C), (n, png_deflate_max_size(W, H, C))
    return out[:n]


def encode_png_gpu(image: torch.Tensor, planar: bool = False, out: "torch.Tensor | None" = None) -> torch.Tensor:
    """uint8 GPU image -- interleaved ``[H,W,C]`` or, ``planar``, ``[C,H,W]`` (what ``pack_rgba8`` leaves); C = 3 or 4 -- to the
    bytes of its PNG file, a uint8 GPU tensor (``gsr_png_encode``).  ``out``: a 16-byte aligned uint8 buffer of at least
    ``png_room(W, H, C)`` bytes to encode into (a slice of a staging buffer); the returned tensor is its first ``png_size`` bytes."""
    import ctypes
    from . import _lib
    if not (image.is_cuda and image.dtype == torch.uint8 and image.dim() == 3):
        raise ValueError("encode_png_gpu expects a uint8 GPU tensor [H,W,C] or [C,H,W]")
    img = image.contiguous()
    C, H, W = (int(v) for v in (img.shape if planar else (img.shape[2], img.shape[0], img.shape[1])))
    n, room = png_size(W, H, C), png_room(W, H, C)
    if out is None:
        out = torch.empty(room, dtype=torch.uint8, device=img.device)
    if not (out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous() and out.numel() >= room and out.data_ptr() % 16 == 0):
        raise ValueError("encode_png_gpu: out must be a contiguous, 16-byte aligned uint8 GPU buffer of png_room(W, H, C) bytes")
    with torch.cuda.device(img.device):
        rc = _lib.lib.gsr_png_encode(img.data_ptr(), W, H, C, 1 if planar else 0, out.data_ptr(),
                                     ctypes.c_void_p(torch.cuda.current_stream(img.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gsr_png_encode failed ({rc}): {_lib.last_error()}")
    return out[:n]


# Staging buffers of closed writers, kept for the next one of the same shape: page-locking 8 x 7 MB of host memory (and the device
# buffers behind it) costs tens of milliseconds -- a sixth of a 400-frame trajectory at 960x540 -- and the frame loop opens a writer
# per call.  Keyed by (device, H, W, compressed, slots); ``release_cached_slots()`` gives the memory back.
_SLOT_CACHE: dict = {}


def release_cached_slots() -> None:
    _SLOT_CACHE.clear()


class GpuFrameWriter:
    """The reference's four files per frame with the file images built ON THE GPU.

    ``submit(name, result)`` queues, on the current stream and with ONE library call (``gsr_frame_files``): the RGBA quantisation
    (save_image's rounding), the turbo-coloured depth preview and the normal map's bytes (the formulas of ``_frame_to_host`` /
    ``depth2img``, in the same fp32 operations), three PNG encodes and the depth plane behind its constant .npy header -- all
    into ONE staging buffer -- then one device-to-host copy of that buffer into a pinned slot and an event.  It returns without waiting for the
    GPU; a host thread waits for the event and writes the four byte ranges to their files.  ``slots`` frames may be in flight
    (the submit of frame i + slots waits for frame i's files).  ``close()`` waits for everything and re-raises the first error.
    The pixels any PNG reader gets, and ``np.load`` of the depth file, are bit-identical to ``FrameWriter``'s and the
    reference's (tests/test_frame_io.py)."""

    def __init__(self, out_dir: str, workers: int = 4, slots: int = 8, deflate: "bool | None" = None):
        from concurrent.futures import ThreadPoolExecutor
        self.out_dir = out_dir
        # compressed PNGs (gsr_frame_files_deflate: what the reference writes, 1.0 - 1.2 x PIL's size) or stored ones (as large as the raw image)
        self.deflate = deflate_default() if deflate is None else bool(deflate)
        self._pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="gpu-frame-writer")
        self._slots, self._n_slots, self._next = [], max(2, slots), 0
        self._shape = None
        self._lut = None
        self._made_dirs: set = set()

    def _prepare(self, H: int, W: int, device):
        """Byte ranges of the four files inside a slot, the staging buffers, the constant .npy header."""
        self._cache_key = (str(device), H, W, bool(self.deflate), self._n_slots)
        cached = _SLOT_CACHE.pop(self._cache_key, None)
        if cached is not None:
            (self._off, self._bytes, self._lengths_at, self._header_len, self._lut, self._slots) = cached
            self._shape = (H, W)
            for slot in self._slots:
                slot["pending"] = None
            return
        size_of, room_of = (png_deflate_max_size, png_deflate_room) if self.deflate else (png_size, png_room)
        sizes = {"images": size_of(W, H, 4), "depth_preview": size_of(W, H, 3), "normal": size_of(W, H, 3)}   # (deflate: upper bounds)
        rooms = {"images": room_of(W, H, 4), "depth_preview": room_of(W, H, 3), "normal": room_of(W, H, 3)}
        header = npy_header((H, W))
        at, off = 0, {}
        for k in ("images", "depth_preview", "normal"):
            off[k] = (at, sizes[k])
            at += (rooms[k] + 15) & ~15
        off["depth"] = (at, len(header) + 4 * H * W)      # (header lengths are multiples of 64: the plane is 4-byte aligned)
        at += (len(header) + 4 * H * W + 15) & ~15
        self._lengths_at = at          # deflate: the three files' lengths (uint64), written by the kernels, copied out with the files
        at += 32
        self._off, self._bytes, self._shape = off, at, (H, W)
        self._lut = torch.from_numpy(TURBO_LUT.copy()).to(device)
        hdr = torch.frombuffer(bytearray(header), dtype=torch.uint8).to(device)
        self._slots = []
        for _ in range(self._n_slots):
            dev = torch.empty(at, dtype=torch.uint8, device=device)
            dev[off["depth"][0]:off["depth"][0] + len(header)] = hdr
            host = torch.empty(at, dtype=torch.uint8, pin_memory=True)
            self._slots.append({"dev": dev, "host": host, "np": host.numpy(), "event": torch.cuda.Event(), "pending": None,
                                "work": torch.empty(10 * H * W, dtype=torch.uint8, device=device),
                                "scratch": (torch.empty(png_deflate_scratch(W, H, 4) + 2 * png_deflate_scratch(W, H, 3), dtype=torch.uint8, device=device)
                                            if self.deflate else None)})
        self._header_len = len(header)
        # the slots are handed out round-robin to whatever stream a frame arrives on: their header fills (queued above on THIS
        # stream) must have landed before another stream copies a slot out.  Once per image size.
        torch.cuda.current_stream(device).synchronize()

    def submit(self, name: str, result: dict) -> None:
        rgba, depth, normal = result["render"], result["depth"].detach(), result["normal"].detach()
        H, W = int(rgba.shape[-2]), int(rgba.shape[-1])
        if self._shape != (H, W):
            self.close(shutdown=False)
            self._prepare(H, W, rgba.device)
        slot = self._slots[self._next]
        self._next = (self._next + 1) % self._n_slots
        if slot["pending"] is not None:
            slot["pending"].result()       # the files of the frame that used this slot are on disk
            slot["pending"] = None
        dev, off = slot["dev"], self._off
        color, alpha = rgba[:3].contiguous(), rgba[3:4].contiguous()
        d, nrm = depth.reshape(H, W).contiguous(), normal.reshape(H, W, 3).contiguous()
        if not all(t.is_cuda and t.dtype == torch.float32 for t in (color, alpha, d, nrm)):
            raise ValueError("GpuFrameWriter.submit expects float32 GPU tensors (a render() result)")
        import ctypes
        from . import _lib
        base = dev.data_ptr()
        with torch.cuda.device(dev.device):
            args = (color.data_ptr(), alpha.data_ptr(), d.data_ptr(), nrm.data_ptr(), 3.0, self._lut.data_ptr(), W, H,
                    base + off["images"][0], base + off["depth_preview"][0], base + off["normal"][0],
                    base + off["depth"][0] + self._header_len, slot["work"].data_ptr())
            stream_ptr = ctypes.c_void_p(torch.cuda.current_stream(dev.device).cuda_stream)
            if self.deflate:
                rc = _lib.lib.gsr_frame_files_deflate(*args, slot["scratch"].data_ptr(), base + self._lengths_at, stream_ptr)
            else:
                rc = _lib.lib.gsr_frame_files(*args, stream_ptr)
            if rc != 0:
                raise RuntimeError(f"gsr_frame_files failed ({rc}): {_lib.last_error()}")
            # (inside the device guard: the copy goes to the current stream of the FRAME's device, and the event is recorded on that
            # stream explicitly -- Event.record() without an argument would take the process's current device)
            slot["host"].copy_(dev, non_blocking=True)
            slot["event"].record(torch.cuda.current_stream(dev.device))
        paths = _frame_paths(self.out_dir, name, self._made_dirs)
        slot["pending"] = self._pool.submit(self._write, slot, paths, dict(off), self._lengths_at if self.deflate else None)

    @staticmethod
    def _write(slot, paths, off, lengths_at=None):
        slot["event"].synchronize()
        buf = slot["np"]
        if lengths_at is not None:     # the compressed files' lengths arrived with them
            got = np.frombuffer(buf[lengths_at:lengths_at + 24], dtype=np.uint64)
            for k, n in zip(("images", "depth_preview", "normal"), got):
                if not 0 < int(n) <= off[k][1]:
                    raise RuntimeError(f"compressed PNG {k!r}: length {int(n)} outside (0, {off[k][1]}]")
                off[k] = (off[k][0], int(n))
        for k, (at, n) in off.items():
            with _open_for_write(paths[k]) as f:
                f.write(memoryview(buf[at:at + n]))
        return paths

    def close(self, shutdown: bool = True) -> None:
        first = None
        for slot in self._slots:
            if slot["pending"] is not None:
                try:
                    slot["pending"].result()
                except Exception as e:   # keep draining: every slot's frame must be off the GPU before the buffers go
                    first = first or e
                slot["pending"] = None
        if self._slots and first is None and getattr(self, "_cache_key", None) is not None:
            # every frame is on disk: the buffers can serve the next writer of this shape (or this one, after a change of size)
            _SLOT_CACHE[self._cache_key] = (self._off, self._bytes, self._lengths_at, self._header_len, self._lut, self._slots)
            self._slots, self._shape = [], None
        if shutdown:
            self._pool.shutdown(wait=True)
        if first is not None:
            raise first

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


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
        self._made_dirs: set = set()
        self._max_pending = max_pending if max_pending > 0 else 4 * workers   # bounds the host memory held by queued frames

    def submit(self, name: str, result: dict) -> None:
        arrays = _frame_to_host(result)
        paths = _frame_paths(self.out_dir, name, self._made_dirs)
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
