"""autovfx_amd.layer_io + csrc/gsr_layerio.hip: Blender's layers from file to GPU memory (blender/blend_all.py:56-75,185-205 -- load_rgb /
load_depth_exr on six PNGs and four EXRs per frame).  The host inflates, the GPU undoes the image predictors; the result must be the
array the reference's loader returns, bit for bit.

CPU: the container parsing -- which files the kernels are given, which go to Pillow / autovfx_amd.exr -- and the parsed stream against a
numpy restatement of the PNG filters.  GPU: the kernels against Pillow / the numpy readers on files with every filter type, image sizes
on both sides of the 64-row band and 4-pixel group boundaries, RGB and RGBA, and the EXR modes."""
import io
import os
import struct
import zlib

import numpy as np
import pytest
import torch
from PIL import Image

from autovfx_amd import exr, frame_io, layer_io


# ---- a PNG written from the specification, with chosen filter types per row ----------------------------------------------------

def _filter_rows(pixels: np.ndarray, types) -> bytes:
    """PNG specification section 9: the FORWARD filters (each byte minus its prediction from the unfiltered neighbours)."""
    h, w, c = pixels.shape
    p = pixels.astype(np.int32)
    left = np.zeros_like(p); left[:, 1:] = p[:, :-1]
    up = np.zeros_like(p); up[1:] = p[:-1]
    corner = np.zeros_like(p); corner[1:, 1:] = p[:-1, :-1]
    est = left + up - corner
    pa, pb, pc = np.abs(est - left), np.abs(est - up), np.abs(est - corner)
    paeth = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, corner))
    pred = {0: np.zeros_like(p), 1: left, 2: up, 3: (left + up) >> 1, 4: paeth}
    rows = []
    for y in range(h):
        t = int(types[y])
        rows.append(bytes([t]) + ((p[y] - pred.get(t, pred[0])[y]) & 255).astype(np.uint8).tobytes())
    return b"".join(rows)


def _png_file(pixels: np.ndarray, types, level=6, idat_pieces=1, extra=b"", interlace=0, depth=8, colour=None) -> bytes:
    h, w, c = pixels.shape
    chunk = lambda kind, data: struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))
    z = zlib.compress(_filter_rows(pixels, types), level)
    cuts = [len(z) * k // idat_pieces for k in range(idat_pieces + 1)]
    colour = (2 if c == 3 else 6) if colour is None else colour
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, interlace)) + extra
            + b"".join(chunk(b"IDAT", z[a:b]) for a, b in zip(cuts, cuts[1:])) + chunk(b"IEND", b""))


def _noise(h, w, c, seed):
    g = np.random.default_rng(seed)
    img = g.integers(0, 256, (h, w, c)).astype(np.uint8)
    img[: h // 3] = (img[: h // 3] // 64) * 64                   # flat stretches and ties between the Paeth candidates
    return img


def test_written_png_is_what_pillow_reads():
    """The test's own PNG writer (every filter type) against Pillow and against the package's numpy reader: the fixture is sound."""
    img = _noise(23, 17, 4, 0)
    data = _png_file(img, [y % 5 for y in range(23)], idat_pieces=3)
    np.testing.assert_array_equal(np.array(Image.open(io.BytesIO(data))), img)
    np.testing.assert_array_equal(frame_io.decode_png(data), img)


@pytest.mark.parametrize("c", [3, 4])
def test_png_scanlines_parses_what_the_kernel_covers(c):
    img = _noise(19, 31, c, c)
    types = [(3 * y + 1) % 5 for y in range(19)]
    w, h, cc, raw = layer_io.png_scanlines(_png_file(img, types, idat_pieces=4, extra=struct.pack(">I", 4) + b"gAMA" + struct.pack(">I", 45455)
                                                     + struct.pack(">I", zlib.crc32(b"gAMA" + struct.pack(">I", 45455)))))
    assert (w, h, cc) == (31, 19, c) and raw == _filter_rows(img, types)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="PNG")
    w, h, cc, raw = layer_io.png_scanlines(buf.getvalue())              # Pillow's adaptive filters
    assert (w, h, cc) == (31, 19, c) and len(raw) == 19 * (1 + 31 * c)


def test_png_flavours_left_to_pillow():
    img = _noise(8, 8, 3, 1)
    ok = _png_file(img, [0] * 8)
    assert layer_io.png_scanlines(ok) is not None
    assert layer_io.png_scanlines(_png_file(img, [0] * 8, interlace=1)) is None
    assert layer_io.png_scanlines(_png_file(img, [0] * 8, depth=16)) is None
    assert layer_io.png_scanlines(_png_file(img, [0] * 8, colour=0)) is None          # grey
    trns = struct.pack(">I", 6) + b"tRNS" + bytes(6) + struct.pack(">I", zlib.crc32(b"tRNS" + bytes(6)))
    assert layer_io.png_scanlines(_png_file(img, [0] * 8, extra=trns)) is None        # a transparent colour: convert("RGBA") applies it
    assert layer_io.png_scanlines(_png_file(img, [0, 1, 2, 3, 4, 5, 0, 0])) is None   # filter type 5 does not exist
    assert layer_io.png_scanlines(ok[:40] + bytes([ok[40] ^ 1]) + ok[41:]) is None    # a flipped bit: the chunk's CRC
    assert layer_io.png_scanlines(ok[:-30]) is None                                    # truncated
    assert layer_io.png_scanlines(b"GIF89a" + ok) is None
    assert layer_io.png_scanlines(_png_file(_noise(2, 4097, 3, 2), [1, 4])) is None   # wider than the kernel's LDS rows: 4096
    for mode in ("P", "L", "LA", "I;16"):
        buf = io.BytesIO()
        Image.fromarray(img).convert(mode).save(buf, format="PNG")
        assert layer_io.png_scanlines(buf.getvalue()) is None, mode


def test_exr_blocks_parses_what_the_kernel_covers(tmp_path):
    g = np.random.default_rng(5)
    z = np.linspace(0.5, 9.0, 40 * 24, dtype=np.float32).reshape(40, 24)
    p = str(tmp_path / "a.exr")
    exr.write_exr(p, {"R": z, "G": z, "B": z + 1, "A": np.ones_like(z)}, compression="ZIP", half=True)
    h, pick, pieces = layer_io.exr_blocks(open(p, "rb").read())
    assert pick == "B" and [len(x) for x in pieces] == [16 * 24 * 8, 16 * 24 * 8, 8 * 24 * 8] and h["bytes_per_line"] == 24 * 8
    exr.write_exr(p, {"Z": np.repeat(z, 8, axis=1)}, compression="ZIPS", half=True)     # (a line per block: long enough to shrink)
    assert layer_io.exr_blocks(open(p, "rb").read())[1] == "Z"
    exr.write_exr(p, {"B": z}, compression="NONE")
    assert layer_io.exr_blocks(open(p, "rb").read()) is None                         # nothing to undo: the host reader
    noise = g.random((16, 24)).astype(np.float32)
    exr.write_exr(p, {"B": noise}, compression="ZIP")                                  # noise does not shrink: the block is stored as it is
    assert layer_io.exr_blocks(open(p, "rb").read()) is None
    np.testing.assert_array_equal(exr.load_depth_exr(p), noise)
    assert layer_io.exr_blocks(b"not an exr file at all") is None


def test_staging_arena_and_inflate_into():
    """The host-side hand-over: zlib streams inflate straight into the arena (libz through ctypes, or Python's zlib and one copy); an
    arena grows without moving what it has handed out and starts over on reset."""
    st = layer_io.Staging()
    payload = bytes(np.random.default_rng(0).integers(0, 7, 300000).astype(np.uint8))
    a = st.take(len(payload))
    assert layer_io.inflate_into(a, zlib.compress(payload, 1)) and bytes(a.numpy()) == payload
    b = st.take(3 << 20)                                     # larger than what is left: a new block; `a` is untouched
    b.zero_()
    assert bytes(a.numpy()) == payload and a.data_ptr() % 64 == 0 and b.data_ptr() % 64 == 0
    assert not layer_io.inflate_into(st.take(len(payload) - 1), zlib.compress(payload))     # does not fit
    assert not layer_io.inflate_into(st.take(len(payload) + 1), zlib.compress(payload))     # does not fill
    assert not layer_io.inflate_into(st.take(len(payload)), zlib.compress(payload)[:-9])    # truncated stream
    assert not layer_io.inflate_into(st.take(16), b"not a zlib stream")
    st.reset()
    c = st.take(len(payload))
    assert layer_io.inflate_into(c, zlib.compress(payload, 9)) and bytes(c.numpy()) == payload
    saved, layer_io._libz = layer_io._libz, None             # the fall-back without a shared zlib
    try:
        d = st.take(len(payload))
        assert layer_io.inflate_into(d, zlib.compress(payload)) and bytes(d.numpy()) == payload
        assert not layer_io.inflate_into(st.take(5), zlib.compress(payload))
    finally:
        layer_io._libz = saved


# ---- the native file readers (csrc/gsr_layerfiles.hip) against the Python restatement: host code, no GPU ----------------------------

def _native_png(buf):
    got = layer_io.read_png_scanlines(buf, layer_io.Staging())
    return None if got is None else (got[1], got[2], got[3], bytes(got[0].numpy()))


def _native_exr(buf, want=None):
    got = layer_io.read_exr_blocks(buf, layer_io.Staging(), want)
    if got is None:
        return None
    host, L = got
    return ({k: getattr(L, k) for k in ("width", "height", "bytes_per_line", "lines_per_block", "channel_at", "channel_bytes")},
            L.channel.decode("latin-1"), bool(L.channel_is_half), bytes(host.numpy()))


def _python_exr(buf, want=None):
    got = layer_io.exr_blocks(buf, want)
    if got is None:
        return None
    L, pick, pieces = got
    return ({k: L[k] for k in ("width", "height", "bytes_per_line", "lines_per_block", "channel_at", "channel_bytes")}, pick,
            L["channel_dtype"] == torch.float16, b"".join(pieces))


def _png_corpus():
    img3, img4 = _noise(8, 8, 3, 1), _noise(19, 31, 4, 2)
    ok = _png_file(img3, [0] * 8)
    trns = struct.pack(">I", 6) + b"tRNS" + bytes(6) + struct.pack(">I", zlib.crc32(b"tRNS" + bytes(6)))
    files = [ok, _png_file(img4, [(3 * y + 1) % 5 for y in range(19)], idat_pieces=5), _png_file(img4, [4] * 19, level=0),
             _png_file(img3, [0] * 8, interlace=1), _png_file(img3, [0] * 8, depth=16), _png_file(img3, [0] * 8, colour=0),
             _png_file(img3, [0] * 8, extra=trns), _png_file(img3, [0, 1, 2, 3, 4, 5, 0, 0]), ok[:40] + bytes([ok[40] ^ 1]) + ok[41:], ok[:-30],
             b"GIF89a" + ok, _png_file(_noise(2, 4097, 3, 2), [1, 4]), _png_file(_noise(2, 4096, 3, 2), [1, 4]), b"", b"\x89PNG\r\n\x1a\n",
             ok + b"trailing bytes after IEND"]
    for mode in ("RGB", "RGBA", "P", "L", "LA", "I;16"):
        buf = io.BytesIO()
        Image.fromarray(img4).convert(mode).save(buf, format="PNG")
        files.append(buf.getvalue())
    # the stream holds more, or less, than the header's image
    chunk = lambda kind, data: struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))
    for rows in (7, 9):
        z = zlib.compress(_filter_rows(_noise(rows, 8, 3, 3), [1] * rows))
        files.append(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 8, 2, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b""))
    return files


def test_native_png_reader_is_the_python_one():
    """Same decision (covered / left to Pillow) and the same scanline bytes for every file of a corpus of good, odd and damaged PNGs, and
    for every truncation and a sweep of single-byte corruptions of a good one."""
    for k, buf in enumerate(_png_corpus()):
        assert _native_png(buf) == layer_io.png_scanlines(buf), f"corpus file {k}"
    good = _png_file(_noise(19, 31, 4, 2), [(3 * y + 1) % 5 for y in range(19)], idat_pieces=3)
    assert _native_png(good) is not None
    for cut in range(0, len(good), 7):
        assert _native_png(good[:cut]) == layer_io.png_scanlines(good[:cut]), f"cut at {cut}"
    g = np.random.default_rng(0)
    for _ in range(300):
        at = int(g.integers(0, len(good)))
        bad = good[:at] + bytes([good[at] ^ (1 << int(g.integers(0, 8)))]) + good[at + 1:]
        assert _native_png(bad) == layer_io.png_scanlines(bad), f"bit flipped in byte {at}"


def _exr_corpus(tmp_path):
    from test_exr import _hand_made
    g = np.random.default_rng(5)
    z = np.linspace(0.5, 9.0, 40 * 24, dtype=np.float32).reshape(40, 24)
    files = []

    def written(channels, **kw):
        p = str(tmp_path / "f.exr")
        exr.write_exr(p, channels, **kw)
        return open(p, "rb").read()
    files.append(written({"R": z, "G": z, "B": z + 1, "A": np.ones_like(z)}, compression="ZIP", half=True))
    files.append(written({"R": z, "G": z, "B": z + 1, "A": np.ones_like(z)}, compression="ZIP", half=False, line_order_decreasing=True))
    files.append(written({"Z": np.repeat(z, 8, axis=1)}, compression="ZIPS", half=True))
    files.append(written({"V": np.repeat(z, 8, axis=1), "W": np.repeat(z, 8, axis=1)}, compression="ZIPS"))
    files.append(written({"Q": np.repeat(z, 8, axis=1), "S": np.repeat(z, 8, axis=1)}, compression="ZIP"))       # none of B G R Y Z V: the first
    files.append(written({"B": z}, compression="NONE"))
    files.append(written({"B": g.random((16, 24)).astype(np.float32)}, compression="ZIP"))                          # stored block
    files += [_hand_made(W=300, H=5, compressed="rle")[0], _hand_made(W=300, H=5, compressed=True)[0], _hand_made()[0]]
    files += [b"not an exr file at all", b"", struct.pack("<ii", 20000630, 2), struct.pack("<ii", 20000630, 2 | 0x200) + files[0][8:]]
    return files


def test_native_exr_reader_is_the_python_one(tmp_path):
    files = _exr_corpus(tmp_path)
    for k, buf in enumerate(files):
        assert _native_exr(buf) == _python_exr(buf), f"corpus file {k}"
        for want in ("A", "G", "Z", "nope"):
            assert _native_exr(buf, want) == _python_exr(buf, want), f"corpus file {k}, channel {want}"
    good = files[0]
    assert _native_exr(good) is not None and _native_exr(good)[1] == "B"
    for cut in list(range(0, 700, 5)) + list(range(700, len(good), 97)):
        assert _native_exr(good[:cut]) == _python_exr(good[:cut]), f"cut at {cut}"
    g = np.random.default_rng(1)
    for _ in range(300):
        at = int(g.integers(0, len(good)))
        bad = good[:at] + bytes([good[at] ^ (1 << int(g.integers(0, 8)))]) + good[at + 1:]
        assert _native_exr(bad) == _python_exr(bad), f"bit flipped in byte {at}"
    for k in (0, 7):                                       # every bit of the header, the offset table and the first block's head
        for at in range(min(len(files[k]), 480)):
            for bit in range(8):
                bad = files[k][:at] + bytes([files[k][at] ^ (1 << bit)]) + files[k][at + 1:]
                assert _native_exr(bad) == _python_exr(bad), f"file {k}: bit {bit} of byte {at} flipped"


# ---- the wave's zlib decoder (csrc/gsr_inflate_core.h): its one-lane host instantiation against zlib ------------------------------------

def _zlib_corpus(sizes=(0, 1, 2, 5, 100, 4095, 4096, 4097, 70000), levels=(0, 1, 6, 9), windows=(15, 9)):
    """(stream, its data): stored / fixed / dynamic blocks, long and overlapping matches, literals only, small windows."""
    g = np.random.default_rng(0)
    out = []
    for size in sizes:
        sparse = np.zeros(size, np.uint8)
        if size:
            sparse[g.integers(0, size, size // 50)] = g.integers(0, 256, size // 50)
        for data in (bytes(size), bytes(g.integers(0, 256, size, dtype=np.uint8)), (b"the quick brown fox jumps over the lazy dog. " * (size // 40 + 1))[:size],
                     bytes((np.arange(size) % 251).astype(np.uint8)), bytes(sparse)):
            for level in levels:
                for strategy in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY):
                    for wbits in windows:
                        c = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
                        out.append((c.compress(data) + c.flush(), data))
    data = bytes((np.arange(50000) % 97).astype(np.uint8)) + bytes(g.integers(0, 256, 20000, dtype=np.uint8))
    for mode in (zlib.Z_SYNC_FLUSH, zlib.Z_FULL_FLUSH):           # empty stored blocks between the others
        c, stream = zlib.compressobj(6), b""
        for k in range(0, len(data), 7001):
            stream += c.compress(data[k:k + 7001]) + c.flush(mode)
        out.append((stream + c.flush(), data))
    return out


def _host_lane_inflate(stream, n):
    import ctypes
    out = ctypes.create_string_buffer(max(n, 1))
    rc = layer_io._lib.lib.gsr_selftest_inflate_host(stream, len(stream), out, n)
    return rc, out.raw[:n]


def _zlib_says(stream, n):
    d = zlib.decompressobj()
    try:
        out = d.decompress(stream)
    except zlib.error:
        return None
    return out if d.eof and len(out) == n else None


def test_wave_inflate_on_one_host_lane_is_zlib():
    """The decoder the GPU runs, instantiated for one host lane: every stream of the corpus inflates to its data; wrong sizes, truncated
    streams are refused; a flipped bit is either refused or zlib accepts the same bytes -- and nothing zlib accepts is refused."""
    for k, (stream, data) in enumerate(_zlib_corpus()):
        assert _host_lane_inflate(stream, len(data)) == (0, data), f"stream {k}: {len(data)} bytes"
    stream, data = _zlib_corpus(sizes=(70000,), levels=(6,), windows=(15,))[8]
    assert _host_lane_inflate(stream, len(data) - 1)[0] == 7 and _host_lane_inflate(stream, len(data) + 1)[0] == 10      # overflow / short output
    assert _host_lane_inflate(stream + b"trailing", len(data)) == (0, data)
    for cut in list(range(0, 40)) + list(range(40, len(stream), max(1, len(stream) // 150))):
        assert _host_lane_inflate(stream[:cut], len(data))[0] != 0, f"truncated at {cut}"
    g = np.random.default_rng(1)
    small = bytes((np.arange(3000) % 13).astype(np.uint8)) + bytes(g.integers(0, 256, 1500, dtype=np.uint8)) + b"abcabcabc" * 100
    for level, strategy in ((6, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_FIXED), (0, zlib.Z_DEFAULT_STRATEGY)):
        c = zlib.compressobj(level, zlib.DEFLATED, 15, 8, strategy)
        stream = c.compress(small) + c.flush()
        for at in range(0, len(stream), 1 if level else 7):
            for bit in range(8):
                bad = stream[:at] + bytes([stream[at] ^ (1 << bit)]) + stream[at + 1:]
                rc, out = _host_lane_inflate(bad, len(small))
                want = _zlib_says(bad, len(small))
                assert (rc == 0) == (want is not None) and (rc != 0 or out == want), f"level {level}: bit {bit} of byte {at}"


# ---- the kernels -----------------------------------------------------------------------------------------------------------------

def _unfilter_on_gpu(data: bytes) -> np.ndarray:
    parsed = layer_io.png_scanlines(data)
    assert parsed is not None
    w, h, c, raw = parsed
    return layer_io.unfilter_png(raw, w, h, c, torch.device("cuda", 0)).cpu().numpy()


def _rgba(img):
    return img if img.shape[2] == 4 else np.concatenate([img, np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=2)


@pytest.mark.gpu
@pytest.mark.parametrize("c", [3, 4])
@pytest.mark.parametrize("hw", [(1, 1), (1, 9), (9, 1), (3, 5), (64, 64), (65, 130), (63, 3), (130, 61), (257, 300), (300, 7), (7, 1100)])
def test_unfilter_every_filter_type(hw, c):
    """Rows of all five types in a fixed rotation and in random order, on images that cross the 64-row bands (1, 2, 3 and 5 of them:
    one and two rounds of the four waves) and the 4-pixel groups."""
    h, w = hw
    img = _noise(h, w, c, h * 1000 + w + c)
    for types in ([y % 5 for y in range(h)], np.random.default_rng(w).integers(0, 5, h), [4] * h, [3] * h):
        got = _unfilter_on_gpu(_png_file(img, types, level=1))
        np.testing.assert_array_equal(got, _rgba(img), err_msg=f"{hw} x {c}, types {list(types)[:8]}...")


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(1920, 1080), (960, 540), (4096, 70)])
def test_unfilter_layer_sized_images_as_pillow_writes_them(size):
    """Blender-sized layers with Pillow's adaptive filter choice: a smooth picture (Sub / Up / Paeth rows), noise, and transparency."""
    w, h = size
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    smooth = np.stack([(127 + 120 * np.sin(xx * 0.01 + k) * np.cos(yy * 0.013 - k)) for k in range(4)], axis=2).astype(np.uint8)
    smooth[..., 3] = np.clip(400 - np.hypot(xx - w / 2, yy - h / 2), 0, 255).astype(np.uint8)
    for img in (smooth, _noise(h, w, 4, 9), smooth[..., :3].copy()):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, format="PNG", compress_level=1)
        got = _unfilter_on_gpu(buf.getvalue())
        np.testing.assert_array_equal(got, np.array(Image.open(io.BytesIO(buf.getvalue())).convert("RGBA")))


@pytest.mark.gpu
def test_load_rgba_is_load_rgb(tmp_path):
    """``load_rgba`` against ``blend_all.load_rgb``'s expression for a covered file, for flavours left to Pillow, and for a missing file."""
    dev = torch.device("cuda", 0)
    img = _noise(70, 90, 4, 3)
    cases = {"rgba.png": Image.fromarray(img), "rgb.png": Image.fromarray(img[..., :3].copy()), "grey.png": Image.fromarray(img[..., 0].copy()),
             "palette.png": Image.fromarray(img[..., :3].copy()).convert("P"), "wide.png": Image.fromarray(_noise(3, 4100, 4, 8))}   # (wider than the kernel takes)
    for name, im in cases.items():
        p = str(tmp_path / name)
        im.save(p)
        got = layer_io.load_rgba(p, dev)
        np.testing.assert_array_equal(got.cpu().numpy(), np.array(Image.open(p).convert("RGBA")), err_msg=name)
    assert layer_io.load_rgba(str(tmp_path / "missing.png"), dev) is None


@pytest.mark.gpu
def test_load_rgba_many_is_one_batch_of_mixed_images(tmp_path):
    """Eleven files of different sizes and channel counts (more than one launch holds), a palette image and a missing file among
    them, on a side stream with a shared staging arena -- the way a ``blend_frames`` pool thread calls it."""
    dev = torch.device("cuda", 0)
    sizes = [(70, 90, 4), (1, 1, 3), (200, 33, 3), (64, 64, 4), (129, 257, 4), (5, 700, 3), (300, 300, 4), (66, 2, 4), (17, 17, 3), (90, 70, 4)]
    paths = []
    for k, (h, w, c) in enumerate(sizes):
        p = str(tmp_path / f"{k}.png")
        Image.fromarray(_noise(h, w, c, k)).save(p, compress_level=1 + k % 3)
        paths.append(p)
    pal = str(tmp_path / "pal.png")
    Image.fromarray(_noise(40, 40, 3, 99)).convert("P").save(pal)
    paths = paths[:4] + [pal, str(tmp_path / "missing.png")] + paths[4:]
    staging = layer_io.Staging()
    stream = torch.cuda.Stream(device=dev)
    for _ in range(2):                                    # the second pass reuses the arena
        staging.reset()
        with torch.cuda.stream(stream):
            got = layer_io.load_rgba_many(paths, dev, staging)
        stream.synchronize()
        for p, g in zip(paths, got):
            if not os.path.exists(p):
                assert g is None
            else:
                np.testing.assert_array_equal(g.cpu().numpy(), np.array(Image.open(p).convert("RGBA")), err_msg=p)


@pytest.mark.gpu
@pytest.mark.parametrize("compression", ["ZIPS", "ZIP"])
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("shape", [(1, 2), (16, 7), (17, 33), (54, 96), (100, 3), (540, 960)])
def test_load_depth_is_load_depth_exr(tmp_path, compression, half, shape):
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(shape[0] + shape[1])
    yy, xx = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float32)
    depth = (3.0 + np.sin(xx * 0.05) + 0.01 * g.random(shape)).astype(np.float32)
    depth[: shape[0] // 2] = 65504.0 if half else 1e10
    p = str(tmp_path / "Image0001.exr")
    exr.write_exr(p, {"R": depth, "G": depth, "B": depth + 1, "A": np.ones(shape, np.float32)}, compression=compression, half=half, level=1)
    want = exr.load_depth_exr(p)
    got = layer_io.load_depth(p, dev)
    if layer_io.exr_blocks(open(p, "rb").read()) is not None:
        assert got.dtype == (torch.float16 if half else torch.float32)
    np.testing.assert_array_equal(got.to(torch.float32).cpu().numpy(), want)
    exr.write_exr(p, {"Z": depth}, compression=compression, half=half)                 # a single channel; a line is one stretch
    got = layer_io.load_depth(p, dev)
    np.testing.assert_array_equal(got.to(torch.float32).cpu().numpy(), exr.load_depth_exr(p))


@pytest.mark.gpu
def test_zlib_streams_inflated_on_the_gpu():
    """``gsr_inflate_zlib_blocks``: the whole corpus in one launch, good streams beside damaged ones -- each good one inflates to its
    data, each damaged one is refused with the host lane's verdict, and no stream disturbs its neighbours."""
    dev = torch.device("cuda", 0)
    corpus = _zlib_corpus(sizes=(0, 1, 5, 100, 4096, 4097, 70000, 300000), levels=(0, 1, 9), windows=(15,))
    streams, sizes, want = [], [], []
    for k, (stream, data) in enumerate(corpus):
        if k % 5 == 3 and len(stream) > 12:                      # damage some: a flipped bit in the middle, or a lost tail
            stream = stream[:len(stream) // 2] + bytes([stream[len(stream) // 2] ^ 16]) + stream[len(stream) // 2 + 1:] if k % 2 else stream[:-5]
        streams.append(stream)
        sizes.append(len(data))
        want.append(_host_lane_inflate(stream, len(data)))
    out, status = layer_io.inflate_zlib_streams(streams, sizes, dev)
    out, status = out.cpu().numpy(), status.cpu().numpy()
    at = refused = 0
    for k, (rc, data) in enumerate(want):
        assert status[k] == rc, f"stream {k}: status {status[k]}, the host lane says {rc}"
        if rc == 0:
            assert bytes(out[at:at + sizes[k]]) == data == corpus[k][1], f"stream {k}"
        else:
            refused += 1
        at += sizes[k]
    assert refused > 10 and refused < len(want) // 3


@pytest.mark.gpu
@pytest.mark.parametrize("half", [False, True])
def test_load_depth_with_the_blocks_inflated_on_the_gpu(tmp_path, half):
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(2)
    yy, xx = np.mgrid[0:270, 0:480].astype(np.float32)
    depth = (3.0 + np.sin(xx * 0.05) + np.cos(yy * 0.03) + 0.001 * g.random((270, 480))).astype(np.float32)
    depth[:100] = 65504.0 if half else 1e10
    p = str(tmp_path / "Image0001.exr")
    for compression in ("ZIP", "ZIPS"):
        exr.write_exr(p, {"R": depth, "G": depth, "B": depth + 1, "A": np.ones_like(depth)}, compression=compression, half=half, level=6)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        got = layer_io.load_depth(p, dev, None, flag)
        assert int(flag.cpu()) == 0 and got.dtype == (torch.float16 if half else torch.float32)
        np.testing.assert_array_equal(got.to(torch.float32).cpu().numpy(), exr.load_depth_exr(p))
    # a damaged block sets the flag (the caller then reads the file on the host, which raises or falls back as before)
    buf = bytearray(open(p, "rb").read())
    first = struct.unpack_from("<Q", buf, exr.read_header(bytes(buf))["offsets_at"])[0]
    buf[first + 8 + 20] ^= 0x40
    open(p, "wb").write(bytes(buf))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    layer_io.load_depth(p, dev, None, flag)
    torch.cuda.synchronize()
    assert int(flag.cpu()) == 1


@pytest.mark.gpu
def test_load_depth_falls_back_for_what_the_kernel_does_not_cover(tmp_path):
    dev = torch.device("cuda", 0)
    g = np.random.default_rng(1)
    noise = g.random((40, 30)).astype(np.float32)             # float noise does not shrink under zlib: blocks stored as they are
    p = str(tmp_path / "Image0001.exr")
    for compression in ("NONE", "ZIP"):
        exr.write_exr(p, {"B": noise, "G": noise}, compression=compression)
        got = layer_io.load_depth(p, dev)
        np.testing.assert_array_equal(got.cpu().numpy(), noise)
    assert layer_io.load_depth(str(tmp_path / "missing.exr"), dev) is None


@pytest.mark.gpu
def test_rle_file_from_the_specification(tmp_path):
    from test_exr import _hand_made
    data, g_plane, z_plane = _hand_made(W=300, H=5, compressed="rle")
    p = str(tmp_path / "rle.exr")
    open(p, "wb").write(data)
    got = layer_io.load_depth(p, torch.device("cuda", 0))
    np.testing.assert_array_equal(got.to(torch.float32).cpu().numpy(), exr.load_depth_exr(p))
