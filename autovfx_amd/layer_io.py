"""Blender's layers from file to GPU memory: the host side of ``csrc/gsr_layerio.hip``.

``blend_all.blend_frames`` (blender/blend_all.py:185-205) reads, per frame, six RGBA PNGs (``load_rgb``: ``Image.open(path).convert
("RGBA")``, :56-60) and four OpenEXR depth passes (``load_depth_exr``: ``cv2.imread(...)[:, :, 0]``, :70-75), all at Blender's
resolution.  Decoding them with Pillow / numpy cost 0.34 s of host time per frame and bound the whole function.  Here the host does
what only a host can do well -- parse the container, run zlib's inflate over each stream (byte-serial; natively, one call per file,
outside the interpreter lock: ``csrc/gsr_layerfiles.hip``) -- and uploads the inflated bytes as they are; the image predictors (PNG's five scanline filters, OpenEXR's byte-wise running
sum and interleave) are undone by kernels (``gsr_png_unfilter``, ``gsr_exr_unpack_channel``).  The results are the arrays the
reference's loaders return, bit for bit; a file the kernels do not cover (16-bit, palette, grey, interlaced or very wide PNGs; PIZ /
tiled / stored-block EXRs) is decoded by Pillow / ``autovfx_amd.exr`` as before.

Everything here runs on the CALLER's current stream.  The host side of a file is: read it, two native calls that parse it and
``inflate`` it straight into page-locked memory (no intermediate ``bytes``, no interpreter lock), one asynchronous copy from there.
The Python parsers in this module (``png_chunks``, ``_exr_plan`` ...) state the same rules readably; the tests hold the two together.  Inflating
into ordinary memory and copying from it was the serial resource of ``blend_frames`` at 8 GB/s: pageable copies go through the
runtime's single staging path.  A ``Staging`` arena belongs to one thread and one stream; ``reset()`` it once that stream has been
synchronised.
"""
import ctypes
import ctypes.util
import os
import struct
import zlib
from typing import Optional

import numpy as np
import torch

from . import _lib, exr

_PNG_SIGNATURE = b"\x89PNG\r\n\x1a\n"

try:
    _libz = ctypes.CDLL(ctypes.util.find_library("z") or "libz.so.1")
    _libz.uncompress.restype = ctypes.c_int
    _libz.uncompress.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulong), ctypes.c_char_p, ctypes.c_ulong]
except OSError:          # no shared zlib to bind: Python's module inflates, one more copy
    _libz = None


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {_lib.last_error()}")


class Staging:
    """Page-locked host memory handed out front to back; ``reset()`` starts over (after the stream that copies from it has drained).
    Grows by allocating a new block: earlier hand-outs stay valid until ``reset``."""

    def __init__(self, nbytes: int = 0):
        self._blocks, self._at, self._room = [], 0, 0
        if nbytes:
            self._grow(nbytes)

    def _grow(self, nbytes):
        # page-locking memory is slow (and every block is locked again when reset() joins them): the first block of a pinned arena
        # holds a frame's worth of 1080p layers, so that a frame usually means ONE allocation
        pinned = torch.cuda.is_available()
        self._blocks.append(torch.empty(max(int(nbytes), (64 << 20) if pinned else (1 << 20)), dtype=torch.uint8, pin_memory=pinned))
        self._at, self._room = 0, self._blocks[-1].numel()

    def take(self, nbytes: int) -> torch.Tensor:
        nbytes = int(nbytes)
        if self._at + nbytes > self._room:
            self._grow(max(nbytes, 2 * self._room))
        out = self._blocks[-1][self._at:self._at + nbytes]
        self._at += (nbytes + 63) & ~63
        return out

    def reset(self):
        if len(self._blocks) > 1:          # one block of the total size from now on
            total = sum(b.numel() for b in self._blocks)
            self._blocks = []
            self._grow(total)
        self._at = 0


def inflate_into(dst: torch.Tensor, data: bytes) -> bool:
    """The zlib stream ``data`` inflated into the host tensor ``dst`` (uint8); False unless it fills it exactly."""
    n = dst.numel()
    if _libz is not None:
        got = ctypes.c_ulong(n)
        rc = _libz.uncompress(ctypes.c_void_p(dst.data_ptr()), ctypes.byref(got), data, len(data))
        return rc == 0 and got.value == n
    try:
        raw = zlib.decompress(data)
    except zlib.error:
        return False
    if len(raw) != n:
        return False
    ctypes.memmove(ctypes.c_void_p(dst.data_ptr()), raw, n)
    return True


def _upload(host: torch.Tensor, device) -> torch.Tensor:
    dst = torch.empty(host.numel(), dtype=torch.uint8, device=device)
    _check(_lib.lib.gsr_upload(ctypes.c_void_p(dst.data_ptr()), ctypes.cast(ctypes.c_void_p(host.data_ptr()), ctypes.c_char_p), host.numel(),
                               _stream_ptr(device)), "gsr_upload")
    return dst


def read_png_scanlines(buf: bytes, staging: "Staging"):
    """``(page-locked scanline stream, w, h, c)`` of a PNG file the unfilter kernel covers, or None: two native calls
    (``gsr_png_file_probe`` / ``gsr_png_file_inflate``: chunk walk, CRCs, zlib's inflate straight into ``staging``, the filter-type
    check), no interpreter lock held meanwhile.  ``png_chunks`` / ``png_scanlines`` below are the same logic in Python."""
    lib = _lib.lib
    info = _lib.PngFileInfo()
    if lib.gsr_png_file_probe(buf, len(buf), ctypes.byref(info)) != 0:
        return None
    host = staging.take(info.scanline_bytes)
    if lib.gsr_png_file_inflate(buf, len(buf), ctypes.c_void_p(host.data_ptr()), info.scanline_bytes) != 0:
        return None
    return host, info.width, info.height, info.channels


def read_exr_blocks(buf: bytes, staging: "Staging", want: Optional[str] = None):
    """``(page-locked inflated blocks, GsrExrFileInfo)`` of an OpenEXR file the unpack kernel covers, or None: two native calls
    (``gsr_exr_file_probe`` / ``gsr_exr_file_inflate``).  ``_exr_plan`` / ``exr_blocks`` below are the same logic in Python."""
    lib = _lib.lib
    info = _lib.ExrFileInfo()
    channel = None if want is None else want.encode()
    if lib.gsr_exr_file_probe(buf, len(buf), channel, ctypes.byref(info)) != 0:
        return None
    host = staging.take(info.blocks_bytes)
    if lib.gsr_exr_file_inflate(buf, len(buf), channel, ctypes.c_void_p(host.data_ptr()), info.blocks_bytes) != 0:
        return None
    return host, info


def png_chunks(buf: bytes):
    """``(width, height, channels, the IDAT chunks' zlib stream)`` of a PNG file the unfilter kernel covers -- 8-bit RGB or RGBA, not
    interlaced, no ``tRNS`` chunk, at most 4096 pixels wide -- or None (another flavour, or a file a real decoder should complain about)."""
    if buf[:8] != _PNG_SIGNATURE:
        return None
    at, ihdr, idat = 8, None, []
    while at + 12 <= len(buf):
        n, = struct.unpack_from(">I", buf, at)
        kind = buf[at + 4:at + 8]
        if at + 12 + n > len(buf):
            return None
        data = buf[at + 8:at + 8 + n]
        if kind in (b"IHDR", b"IDAT"):
            crc, = struct.unpack_from(">I", buf, at + 8 + n)
            if zlib.crc32(data, zlib.crc32(kind)) != crc:
                return None
        if kind == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        elif kind == b"IDAT":
            idat.append(data)
        elif kind in (b"tRNS", b"PLTE"):
            return None
        elif kind == b"IEND":
            break
        at += 12 + n
    if ihdr is None or not idat:
        return None
    w, h, depth, colour, compression, filtering, interlace = ihdr
    if depth != 8 or colour not in (2, 6) or compression or filtering or interlace or w < 1 or h < 1:
        return None
    if _lib.lib.gsr_png_unfilter_scratch(w, h) == 0 or h * (1 + 4 * w) > (1 << 30):      # (too wide for the kernel / a header not to be trusted with staging memory)
        return None
    return w, h, (3 if colour == 2 else 4), (idat[0] if len(idat) == 1 else b"".join(idat))


def png_scanlines(buf: bytes):
    """``(width, height, channels, inflated scanline stream)`` of a covered PNG file, or None: ``png_chunks`` plus what ``load_rgba`` checks
    after inflating (the stream's length, the filter-type bytes)."""
    parsed = png_chunks(buf)
    if parsed is None:
        return None
    w, h, c, stream = parsed
    try:
        raw = zlib.decompress(stream)
    except zlib.error:
        return None
    stride = 1 + w * c
    if len(raw) != h * stride or max(raw[0::stride]) > 4:
        return None
    return w, h, c, raw


def _unfilter_many(jobs, device):
    """``jobs``: ``[(page-locked scanline stream, w, h, c)]`` -> the RGBA8 images, by one ``gsr_png_unfilter_batch`` call (a workgroup per
    image, side by side) on the current stream."""
    lib = _lib.lib
    if not jobs:
        return []
    table = (_lib.PngUnfilterJob * len(jobs))()
    held, outs = [], []
    for k, (host, w, h, c) in enumerate(jobs):
        staged = _upload(host, device)
        out = torch.empty((h, w, 4), dtype=torch.uint8, device=device)
        scratch = torch.empty(lib.gsr_png_unfilter_scratch(w, h), dtype=torch.uint8, device=device)
        table[k] = _lib.PngUnfilterJob(staged.data_ptr(), w, h, c, out.data_ptr(), scratch.data_ptr())
        held += [staged, scratch]         # (freed to the stream's pool after the launch below: ordered behind it)
        outs.append(out)
    with torch.cuda.device(device):
        _check(lib.gsr_png_unfilter_batch(len(jobs), ctypes.byref(table), _stream_ptr(device)), "gsr_png_unfilter_batch")
    return outs


def unfilter_png(raw: bytes, w: int, h: int, c: int, device) -> torch.Tensor:
    """The inflated scanline stream of an 8-bit RGB / RGBA PNG -> ``uint8[h, w, 4]`` on ``device`` (alpha 255 for RGB).  Blocking."""
    host = Staging(len(raw)).take(len(raw))
    ctypes.memmove(ctypes.c_void_p(host.data_ptr()), raw, len(raw))
    out, = _unfilter_many([(host, w, h, c)], device)
    torch.cuda.current_stream(device).synchronize()
    return out


def load_rgba_many(paths, device, staging: Optional[Staging] = None):
    """``blend_all.load_rgb`` (:56-60) for several files with the results on the GPU: per path ``uint8[H, W, 4]`` =
    ``np.array(Image.open(path).convert("RGBA"))``, or None where the file does not exist.  The files the kernel covers are inflated
    into ``staging`` and unfiltered by ONE batch launch; with a ``staging`` arena the call only queues work on the current stream."""
    own = staging is None
    if own:
        staging = Staging()
    results, jobs, slots = [None] * len(paths), [], []
    for k, path in enumerate(paths):
        if not os.path.exists(path):
            continue
        with open(path, "rb") as f:
            buf = f.read()
        job = read_png_scanlines(buf, staging)
        if job is not None:
            jobs.append(job)
            slots.append(k)
            continue
        from PIL import Image       # any other flavour (or a damaged file: Pillow says what is wrong with it), as in the reference
        import io
        results[k] = torch.from_numpy(np.array(Image.open(io.BytesIO(buf)).convert("RGBA"))).to(device)
    for k, out in zip(slots, _unfilter_many(jobs, device)):
        results[k] = out
    if own:
        torch.cuda.current_stream(device).synchronize()
    return results


def load_rgba(path: str, device, staging: Optional[Staging] = None) -> Optional[torch.Tensor]:
    """``load_rgba_many`` for one file."""
    return load_rgba_many([path], device, staging)[0]


def _exr_plan(buf: bytes, want: Optional[str] = None):
    """``(header + layout, channel name, codec, [(compressed block, inflated size)] in increasing y)`` of a scanline OpenEXR file the
    unpack kernel covers -- ZIP / ZIPS / RLE, the wanted channel half or float, every block actually compressed -- or None."""
    try:
        return _exr_plan_unchecked(buf, want)
    except (ValueError, KeyError, IndexError, struct.error, OverflowError, UnicodeDecodeError):      # a damaged header
        return None


def _exr_plan_unchecked(buf, want):
    h = exr.read_header(buf)
    name, lines_per_block = exr._COMPRESSION.get(h["compression"], (None, 0))
    if name not in ("ZIP", "ZIPS", "RLE") or len(h["attributes"]["compression"][1]) != 1:
        return None
    names = [n for n, _p in h["channels"]]
    if not names or any(p not in exr._PIXEL for _n, p in h["channels"]):
        return None
    pick = want if want else next((n for n in ("B", "G", "R", "Y", "Z", "V") if n in names), names[0])
    if pick not in names:
        return None
    xmin, ymin, xmax, ymax = h["data_window"]
    W, H = xmax - xmin + 1, ymax - ymin + 1
    if W < 1 or H < 1 or W > (1 << 20) or H > (1 << 24):
        return None
    bytes_per_line = sum(exr._PIXEL[p].itemsize for _n, p in h["channels"]) * W
    if bytes_per_line * lines_per_block > (1 << 30) or bytes_per_line * H > (1 << 32):
        return None
    c_at, dt = 0, None
    for n, p in h["channels"]:        # (a name listed twice: the first entry, as the native reader takes it)
        if n == pick and dt is None:
            dt, at_pick = exr._PIXEL[p], c_at
        c_at += exr._PIXEL[p].itemsize * W
    if dt.kind != "f":
        return None
    n_blocks = (H + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from(f"<{n_blocks}Q", buf, h["offsets_at"])
    pieces = []
    for k, off in enumerate(offsets):
        if off + 8 > len(buf):
            return None
        y, size = struct.unpack_from("<ii", buf, off)
        expected = min(lines_per_block, H - k * lines_per_block) * bytes_per_line
        if y != ymin + k * lines_per_block or size < 0 or size >= expected or off + 8 + size > len(buf):   # (a block that did not shrink is stored without the predictor)
            return None
        pieces.append((buf[off + 8:off + 8 + size], expected))
    layout = dict(h, width=W, height=H, bytes_per_line=bytes_per_line, lines_per_block=lines_per_block, channel_at=at_pick, channel_bytes=dt.itemsize * W,
                  channel_dtype=torch.float16 if dt.itemsize == 2 else torch.float32)
    return layout, pick, name, pieces


def exr_blocks(buf: bytes, want: Optional[str] = None):
    """``(header + layout, channel name, inflated blocks in increasing y)`` of a covered OpenEXR file, or None."""
    plan = _exr_plan(buf, want)
    if plan is None:
        return None
    layout, pick, name, pieces = plan
    out = []
    for data, expected in pieces:
        try:
            raw = zlib.decompress(data) if name != "RLE" else exr._rle_decode(data, expected)
        except (zlib.error, ValueError):
            return None
        if len(raw) != expected:
            return None
        out.append(raw)
    return layout, pick, out


def exr_inflate_on_gpu() -> bool:
    """Whether ZIP / ZIPS OpenEXR blocks are inflated by the GPU (``AUTOVFX_AMD_EXR_INFLATE=gpu``, the default) or by zlib on the host."""
    return os.environ.get("AUTOVFX_AMD_EXR_INFLATE", "gpu").lower() != "host"


def inflate_zlib_streams(streams, sizes, device):
    """Independent zlib streams inflated on the GPU (``gsr_inflate_zlib_blocks``: a single-wave workgroup each): ``(out uint8[sum(sizes)],
    status int32[n])`` device tensors; stream i must inflate to exactly ``sizes[i]`` bytes, ``status[i]`` says whether it did.  Blocking."""
    n = len(streams)
    jobs = np.zeros((n, 4), np.uint32)
    at = out_at = 0
    for k, (data, size) in enumerate(zip(streams, sizes)):
        jobs[k] = (at, len(data), out_at, size)
        at += (len(data) + 3) & ~3
        out_at += size
    packed = np.zeros(at + 4, np.uint8)
    for k, data in enumerate(streams):
        packed[jobs[k, 0]:jobs[k, 0] + len(data)] = np.frombuffer(data, np.uint8)
    d_packed, d_jobs = torch.from_numpy(packed).to(device), torch.from_numpy(jobs.view(np.int32)).to(device)
    out = torch.empty(max(out_at, 1), dtype=torch.uint8, device=device)
    status = torch.full((max(n, 1),), -1, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _check(_lib.lib.gsr_inflate_zlib_blocks(d_packed.data_ptr(), out.data_ptr(), d_jobs.data_ptr(), n, status.data_ptr(), None, _stream_ptr(device)),
               "gsr_inflate_zlib_blocks")
    torch.cuda.current_stream(device).synchronize()
    return out[:out_at], status[:n]


def _depth_blocks_on_gpu(bufs, device, staging: Staging, error_flag):
    """The inflated blocks of several ZIP / ZIPS OpenEXR files as device tensors, inflated ON the GPU by ONE launch (a single-wave
    workgroup per scanline block of every file): the host only copies the compressed streams together (``gsr_exr_file_pack``).  Per
    file ``(blocks, info)`` or None (not covered: RLE, another codec, a damaged header).  A stream the decoder refuses sets
    ``error_flag`` (device int32[1]) -- the caller looks at it once its stream has drained and reads the files on the host instead."""
    lib = _lib.lib
    infos = []
    for buf in bufs:
        info = _lib.ExrFileInfo()
        ok = buf is not None and lib.gsr_exr_file_probe(buf, len(buf), None, ctypes.byref(info)) == 0 and info.compression != 1
        infos.append(info if ok else None)
    n_jobs = sum(i.n_blocks for i in infos if i is not None)
    if n_jobs == 0:
        return [None] * len(bufs)
    room = sum(len(b) + 4 * i.n_blocks + 4 for b, i in zip(bufs, infos) if i is not None)
    host = staging.take(16 * n_jobs + room)                       # [jobs of every file | their packed streams]: one upload
    jobs = host[:16 * n_jobs].numpy().view(np.uint32).reshape(n_jobs, 4)
    job_at, packed_at, out_at, spans = 0, 0, 0, []
    for k, (buf, info) in enumerate(zip(bufs, infos)):
        if info is None:
            spans.append(None)
            continue
        packed_bytes = ctypes.c_size_t(0)
        if lib.gsr_exr_file_pack(buf, len(buf), None, ctypes.c_void_p(host.data_ptr() + 16 * n_jobs + packed_at), room - packed_at,
                                 ctypes.c_void_p(host.data_ptr() + 16 * job_at), ctypes.byref(packed_bytes)) != 0:
            raise RuntimeError("gsr_exr_file_pack refused a file gsr_exr_file_probe covers")
        mine = jobs[job_at:job_at + info.n_blocks]
        mine[:, 0] += packed_at                                   # the file's offsets -> offsets into the common buffers
        mine[:, 2] += out_at
        spans.append((out_at, info.blocks_bytes))
        job_at += info.n_blocks
        packed_at += (packed_bytes.value + 15) & ~15
        out_at += (info.blocks_bytes + 15) & ~15
    staged = _upload(host[:16 * n_jobs + packed_at + 4], device)
    blocks = torch.empty(out_at, dtype=torch.uint8, device=device)
    status = torch.empty(n_jobs, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _check(lib.gsr_inflate_zlib_blocks(staged.data_ptr() + 16 * n_jobs, blocks.data_ptr(), staged.data_ptr(), n_jobs, status.data_ptr(),
                                           error_flag.data_ptr(), _stream_ptr(device)), "gsr_inflate_zlib_blocks")
    return [None if sp is None else (blocks[sp[0]:sp[0] + sp[1]], info) for sp, info in zip(spans, infos)]


def _unpack_plane(staged: torch.Tensor, L, device) -> torch.Tensor:
    plane = torch.empty((L.height, L.channel_bytes), dtype=torch.uint8, device=device)
    with torch.cuda.device(device):
        _check(_lib.lib.gsr_exr_unpack_channel(staged.data_ptr(), L.height, L.bytes_per_line, L.lines_per_block, L.channel_at, L.channel_bytes,
                                               plane.data_ptr(), _stream_ptr(device)), "gsr_exr_unpack_channel")
    return plane.view(torch.float16 if L.channel_is_half else torch.float32)


def load_depth_many(paths, device, staging: Optional[Staging] = None, error_flag: Optional[torch.Tensor] = None):
    """``blend_all.load_depth_exr`` (:70-75) for several files with the results on the GPU: per path the plane
    ``cv2.imread(path, ANYCOLOR | ANYDEPTH)[:, :, 0]`` returns (the file's B channel) in the file's precision (float16 for Blender's
    half-float passes: widening it to float32 is exact and the caller's), None where the file does not exist.  With a ``staging`` arena
    the call only queues work on the current stream.  With ``error_flag`` (device int32[1], zero) the zlib streams of ZIP / ZIPS files
    are inflated by the GPU, all files' blocks in one launch; the caller checks the flag after synchronising and, if it is set, calls
    again without it (zlib on the host then says what is wrong with the file)."""
    own = staging is None
    if own:
        staging = Staging()
    bufs = []
    for path in paths:
        if os.path.exists(path):
            with open(path, "rb") as f:
                bufs.append(f.read())
        else:
            bufs.append(None)
    on_gpu = _depth_blocks_on_gpu(bufs, device, staging, error_flag) if error_flag is not None else [None] * len(bufs)
    results = []
    for path, buf, got in zip(paths, bufs, on_gpu):
        if buf is None:
            results.append(None)
            continue
        if got is None:
            host_got = read_exr_blocks(buf, staging)
            if host_got is not None:
                got = (_upload(host_got[0], device), host_got[1])
        if got is not None:
            results.append(_unpack_plane(got[0], got[1], device))
        else:
            from . import compositor       # OpenCV where installed, autovfx_amd.exr otherwise: what the kernels do not cover
            results.append(torch.from_numpy(np.ascontiguousarray(compositor.load_depth_exr(path))).to(device))
    if own:
        torch.cuda.current_stream(device).synchronize()
    return results


def load_depth(path: str, device, staging: Optional[Staging] = None, error_flag: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """``load_depth_many`` for one file."""
    return load_depth_many([path], device, staging, error_flag)[0]
