"""A minimal OpenEXR scanline reader (and writer) for Blender's depth passes, so that ``blend_frames`` does not need OpenCV.

The reference loads every Blender depth layer with ``cv2.imread(path, cv2.IMREAD_ANYCOLOR | cv2.IMREAD_ANYDEPTH)[:, :, 0]``
(``blender/blend_all.py:70-75``, needs ``OPENCV_IO_ENABLE_OPENEXR=1``): the file is what Blender's compositor "File Output" node
writes with ``format.file_format = 'OPEN_EXR'`` (``blender/all_rendering.py:274-278``) -- a single-part scanline image, channels
``R, G, B`` (and ``A``) all holding the Z pass, 16-bit half or 32-bit float, ZIP compression by default.  OpenCV hands the
channels back in B, G, R order as float32, so ``[:, :, 0]`` is the file's ``B`` channel.

Supported: single-part scanline files, pixel types HALF / FLOAT / UINT, compressions NONE, RLE, ZIPS (one scanline per block) and
ZIP (16 scanlines per block), any line order, any data window.  Not supported (a clear error names the codec): tiled, deep and
multi-part files, PIZ / PXR24 / B44 / DWA.  File layout and the ZIP pre-processing (byte de-interleaving + delta predictor) follow
the OpenEXR file-layout document; ``tests/test_exr.py`` round-trips through the writer below and, where OpenCV or the ``OpenEXR``
module is importable, compares with them.
"""
from __future__ import annotations

import struct
import zlib
from typing import Dict, Optional, Sequence

import numpy as np

MAGIC = 20000630
_COMPRESSION = {0: ("NONE", 1), 1: ("RLE", 1), 2: ("ZIPS", 1), 3: ("ZIP", 16), 4: ("PIZ", 32), 5: ("PXR24", 16), 6: ("B44", 32),
                7: ("B44A", 32), 8: ("DWAA", 32), 9: ("DWAB", 256)}
_PIXEL = {0: np.dtype("<u4"), 1: np.dtype("<f2"), 2: np.dtype("<f4")}


def _cstr(buf: bytes, at: int):
    end = buf.index(b"\0", at)
    return buf[at:end].decode("latin-1"), end + 1


def _undo_predictor_and_interleave(raw: bytes) -> bytes:
    """The inverse of what ZIP / ZIPS / RLE blocks are pre-processed with: a running byte sum (delta predictor, bias 128), then the
    two halves -- bytes at even offsets, bytes at odd offsets -- woven back together."""
    t = np.frombuffer(raw, np.uint8).copy()
    if t.size == 0:
        return raw
    # t[i] = t[i-1] + d[i] - 128 (mod 256): a running sum in uint8 arithmetic, which wraps the way the format means it (-128 = +128 mod 256).
    # (An int64 running sum of the same bytes was most of a frame's decoding time.)
    t[1:] += np.uint8(128)
    t = np.add.accumulate(t, dtype=np.uint8)
    half = (t.size + 1) // 2
    out = np.empty_like(t)
    out[0::2] = t[:half]
    out[1::2] = t[half:]
    return out.tobytes()


def _undo_for_one_channel(raw: bytes, lines: int, bytes_per_line: int, at: int, width_bytes: int) -> np.ndarray:
    """The bytes ``[at, at + width_bytes)`` of every line of a compressed block -- one channel -- as ``uint8[lines, width_bytes]``, without
    undoing the rest of the block.  The running sum is serial (0.3 GB/s in numpy) and was the largest item of decoding a Blender depth
    pass, of whose four identical channels the compositor wants one: the sum in front of each wanted stretch comes from vectorised row
    totals, the element-by-element sum runs over the wanted stretches alone."""
    t = np.frombuffer(raw, np.uint8).copy()
    t[1:] += np.uint8(128)
    lb2, a2, w2 = bytes_per_line // 2, at // 2, width_bytes // 2
    # The block is [even bytes of all lines | odd bytes of all lines]; each half is `lines` rows of lb2 bytes.
    rows = t.reshape(2 * lines, lb2)
    want = np.add.accumulate(rows[:, a2:a2 + w2], axis=1, dtype=np.uint8)
    before = np.add.reduce(rows[:, :a2], axis=1, dtype=np.uint8)
    after = np.add.reduce(rows[:, a2 + w2:], axis=1, dtype=np.uint8)
    row_total = before + want[:, -1] + after
    carry = np.zeros(2 * lines, np.uint8)
    np.add.accumulate(row_total[:-1], out=carry[1:], dtype=np.uint8)
    want += (carry + before)[:, None]
    out = np.empty((lines, width_bytes), np.uint8)
    out[:, 0::2] = want[:lines]
    out[:, 1::2] = want[lines:]
    return out


def _predictor_and_deinterleave(raw: bytes) -> bytes:
    a = np.frombuffer(raw, np.uint8)
    t = np.concatenate((a[0::2], a[1::2])).astype(np.int64)
    d = t.copy()
    d[1:] = (t[1:] - t[:-1] + 128 + 256) & 255
    return d.astype(np.uint8).tobytes()


def _rle_decode(data: bytes, expected: int) -> bytes:
    out, i = bytearray(), 0
    while i < len(data):
        n = data[i] - 256 if data[i] > 127 else data[i]
        i += 1
        if n < 0:                      # -n literal bytes
            out += data[i:i - n]
            i += -n
        else:                          # the next byte, n + 1 times
            out += data[i:i + 1] * (n + 1)
            i += 1
    if len(out) != expected:
        raise ValueError(f"RLE block decodes to {len(out)} bytes, expected {expected}")
    return bytes(out)


def read_header(buf: bytes):
    magic, version = struct.unpack_from("<ii", buf, 0)
    if magic != MAGIC:
        raise ValueError("not an OpenEXR file (bad magic number)")
    if version & 0xFF != 2:
        raise ValueError(f"OpenEXR file format version {version & 0xFF} is not supported")
    if version & 0x200:
        raise ValueError("tiled OpenEXR files are not supported (Blender's File Output node writes scanline files)")
    if version & 0x1800:
        raise ValueError("deep / multi-part OpenEXR files are not supported")
    at, attrs = 8, {}
    while buf[at] != 0:
        name, at = _cstr(buf, at)
        kind, at = _cstr(buf, at)
        size, = struct.unpack_from("<i", buf, at)
        at += 4
        attrs[name] = (kind, buf[at:at + size])
        at += size
    at += 1
    channels, c = [], 0
    cl = attrs["channels"][1]
    while cl[c] != 0:
        name, c = _cstr(cl, c)
        ptype, _plinear, xs, ys = struct.unpack_from("<iB3xii", cl, c)
        c += 16
        if xs != 1 or ys != 1:
            raise ValueError(f"channel {name!r} is sub-sampled ({xs}x{ys}): not supported")
        channels.append((name, ptype))
    compression = attrs["compression"][1][0]
    xmin, ymin, xmax, ymax = struct.unpack("<4i", attrs["dataWindow"][1])
    line_order = attrs["lineOrder"][1][0] if "lineOrder" in attrs else 0
    return {"channels": channels, "compression": compression, "data_window": (xmin, ymin, xmax, ymax), "line_order": line_order,
            "attributes": attrs, "offsets_at": at}


def read_exr(path_or_bytes, only: Optional[str] = None) -> Dict[str, np.ndarray]:
    """The channels of a scanline OpenEXR file as ``{name: array[H, W]}`` in the file's pixel types (float16 / float32 / uint32); with
    ``only``, that channel alone (and only its share of the decoding work)."""
    buf = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    h = read_header(buf)
    if only is not None and only not in [n for n, _p in h["channels"]]:
        raise KeyError(f"the file has no channel {only!r} (it has {[n for n, _p in h['channels']]})")
    name, lines_per_block = _COMPRESSION.get(h["compression"], (f"#{h['compression']}", 0))
    if name not in ("NONE", "RLE", "ZIPS", "ZIP"):
        raise ValueError(f"OpenEXR compression {name} is not supported by this reader (NONE, RLE, ZIPS and ZIP are; Blender's default is ZIP). "
                         "Re-save with ZIP, or install OpenCV")
    xmin, ymin, xmax, ymax = h["data_window"]
    W, H = xmax - xmin + 1, ymax - ymin + 1
    chans = h["channels"]
    bytes_per_line = sum(_PIXEL[p].itemsize for _n, p in chans) * W
    n_blocks = (H + lines_per_block - 1) // lines_per_block
    offsets = struct.unpack_from(f"<{n_blocks}Q", buf, h["offsets_at"])
    out = {n: np.empty((H, W), _PIXEL[p]) for n, p in chans if only is None or n == only}
    spans, at = {}, 0
    for n, p in chans:
        spans[n] = (at, _PIXEL[p].itemsize * W, _PIXEL[p])
        at += _PIXEL[p].itemsize * W
    for off in offsets:
        y, size = struct.unpack_from("<ii", buf, off)
        data = bytes(buf[off + 8:off + 8 + size])
        y0 = y - ymin
        lines = min(lines_per_block, H - y0)
        expected = lines * bytes_per_line
        if name != "NONE" and size < expected:          # (a block that did not shrink is stored as it is)
            raw = zlib.decompress(data) if name in ("ZIP", "ZIPS") else _rle_decode(data, expected)
            if only is not None and len(raw) == expected:
                c_at, c_bytes, dt = spans[only]
                out[only][y0:y0 + lines] = _undo_for_one_channel(raw, lines, bytes_per_line, c_at, c_bytes).view(dt)
                continue
            data = _undo_predictor_and_interleave(raw)
        if len(data) != expected:
            raise ValueError(f"scanline block at y = {y}: {len(data)} bytes, expected {expected}")
        blk = np.frombuffer(data, np.uint8).reshape(lines, bytes_per_line)     # a line: channel after channel, W values each
        for n in out:
            c_at, c_bytes, dt = spans[n]
            out[n][y0:y0 + lines] = np.ascontiguousarray(blk[:, c_at:c_at + c_bytes]).view(dt)
    return out


def load_depth_exr(path: str) -> Optional[np.ndarray]:
    """What ``cv2.imread(path, IMREAD_ANYCOLOR | IMREAD_ANYDEPTH)[:, :, 0]`` gives for Blender's depth pass (``blend_all.py:70-75``):
    float32 ``[H, W]`` of the file's ``B`` channel (OpenCV orders a colour image B, G, R); a file without colour channels yields its
    first channel (``Y``, ``Z``, ``V``...)."""
    buf = open(path, "rb").read()
    names = [n for n, _p in read_header(buf)["channels"]]
    pick = next((n for n in ("B", "G", "R", "Y", "Z", "V") if n in names), names[0])
    return read_exr(buf, only=pick)[pick].astype(np.float32)


def write_exr(path: str, channels: Dict[str, np.ndarray], compression: str = "ZIP", half: bool = False,
              line_order_decreasing: bool = False, level: int = 6) -> None:
    """A scanline OpenEXR file with the given channels (``[H, W]`` arrays, stored alphabetically by name as the format requires), as
    Blender writes its passes: for tests and for synthetic Blender trees (bench.py).  ``compression``: NONE, ZIPS or ZIP."""
    code = {"NONE": 0, "ZIPS": 2, "ZIP": 3}[compression]
    lines_per_block = _COMPRESSION[code][1]
    names = sorted(channels)
    H, W = channels[names[0]].shape
    ptype = 1 if half else 2
    dt = _PIXEL[ptype]
    planes = {n: np.ascontiguousarray(channels[n], dtype=dt) for n in names}

    def attr(name, kind, data):
        return name.encode() + b"\0" + kind.encode() + b"\0" + struct.pack("<i", len(data)) + data

    chlist = b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", ptype, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, W - 1, H - 1)
    header = (struct.pack("<ii", MAGIC, 2) + attr("channels", "chlist", chlist) + attr("compression", "compression", bytes([code]))
              + attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
              + attr("lineOrder", "lineOrder", bytes([1 if line_order_decreasing else 0]))
              + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0))
              + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0")
    starts = list(range(0, H, lines_per_block))
    if line_order_decreasing:
        starts = starts[::-1]
    blocks = []
    for y0 in starts:
        lines = min(lines_per_block, H - y0)
        raw = b"".join(planes[n][y0 + ln].tobytes() for ln in range(lines) for n in names)
        data = raw
        if code:
            packed = zlib.compress(_predictor_and_deinterleave(raw), level)
            if len(packed) < len(raw):
                data = packed
        blocks.append((y0, struct.pack("<ii", y0, len(data)) + data))
    table_at = len(header)
    at = table_at + 8 * len(blocks)
    by_y = {}
    for y0, blob in blocks:
        by_y[y0] = at
        at += len(blob)
    table = b"".join(struct.pack("<Q", by_y[y0]) for y0 in sorted(by_y))      # (the offset table is in increasing-y order)
    with open(path, "wb") as f:
        f.write(header + table + b"".join(blob for _y, blob in blocks))
