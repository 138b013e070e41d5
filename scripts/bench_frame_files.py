#!/usr/bin/env python
"""The frame-file and compositor-input kernels on their own (gsr_frameio.hip): gsr_frame_files (three PNG encodes + previews + depth
copy), one gsr_png_encode, the two resizes and the composite from 2x layers, each timed with HIP events over many launches and priced
against its algorithmic bytes.  Run under rocprofv3 --kernel-trace --stats for the per-kernel table (profiles/r05_frameio_kernel_stats.csv)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autovfx_amd import compositor, frame_io          # noqa: E402
from autovfx_amd.frame_parallel import pack_rgba8      # noqa: E402

dev = torch.device("cuda", 0)
W, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "960x540").split("x"))
g = torch.Generator(device=dev).manual_seed(0)
HBM = 8000.0


def timed(fn, n=200):
    for _ in range(10):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def row(name, ms, nbytes, what):
    print(json.dumps({"kernel": name, "size": f"{W}x{H}", "ms": round(ms, 5), "alg_bytes": int(nbytes), "GBps": round(nbytes / ms / 1e6, 1),
                      "frac_of_hbm_peak": round(nbytes / ms / 1e6 / HBM, 4), "bytes": what}), flush=True)


# an image with the statistics of a rendered frame (smooth, a little noise): noise alone would not compress -- the compressed
# encoder would send every block stored
yy, xx = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
rgba = torch.stack([0.5 + 0.4 * torch.sin(xx * (0.011 + 0.003 * k) + yy * (0.007 + 0.002 * k) + k) for k in range(4)]) \
    + 0.01 * torch.randn(4, H, W, device=dev, generator=g)
rgba = rgba.clamp(0, 1)
u8 = pack_rgba8(rgba[:3], rgba[3:4])
out = torch.empty(frame_io.png_room(W, H, 4), dtype=torch.uint8, device=dev)
row("gsr_png_encode RGBA (memset + png_encode_kernel + png_finish_kernel)", timed(lambda: frame_io.encode_png_gpu(u8, planar=True, out=out)),
    4 * W * H + frame_io.png_size(W, H, 4), "4 B/pixel in, the file (4 B/pixel + 1 B/row + 5 B/65535) out")
result = {"render": rgba, "depth": torch.rand(H, W, device=dev, generator=g) * 5, "normal": torch.nn.functional.normalize(torch.randn(H, W, 3, device=dev, generator=g), dim=-1)}
import tempfile
with tempfile.TemporaryDirectory() as d:
    w = frame_io.GpuFrameWriter(d, workers=1, slots=2, deflate=False)
    w._prepare(H, W, dev)
    slot = w._slots[0]
    import ctypes
    from autovfx_amd import _lib
    base, off = slot["dev"].data_ptr(), w._off
    color, alpha, dpt, nrm = rgba[:3].contiguous(), rgba[3:4].contiguous(), result["depth"], result["normal"].contiguous()

    def files():
        _lib.lib.gsr_frame_files(color.data_ptr(), alpha.data_ptr(), dpt.data_ptr(), nrm.data_ptr(), 3.0, w._lut.data_ptr(), W, H,
                                 base + off["images"][0], base + off["depth_preview"][0], base + off["normal"][0],
                                 base + off["depth"][0] + w._header_len, slot["work"].data_ptr(),
                                 ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    total_files = sum(n for _a, n in off.values())
    row("gsr_frame_files (pack + previews + 3 PNG encodes + depth copy: 12 launches)", timed(files),
        (16 + 4 + 12) * W * H + 10 * W * H * 2 + total_files, "32 B/pixel of float images in, 10 B/pixel of 8-bit images written and read, the four files out")
    w.close()
    # the same four files with the three PNGs COMPRESSED (gsr_frame_files_deflate: every kernel launched once for the three images)
    w = frame_io.GpuFrameWriter(d, workers=1, slots=2, deflate=True)
    w._prepare(H, W, dev)
    slot = w._slots[0]
    base, off = slot["dev"].data_ptr(), w._off

    def files_deflate():
        _lib.lib.gsr_frame_files_deflate(color.data_ptr(), alpha.data_ptr(), dpt.data_ptr(), nrm.data_ptr(), 3.0, w._lut.data_ptr(), W, H,
                                         base + off["images"][0], base + off["depth_preview"][0], base + off["normal"][0],
                                         base + off["depth"][0] + w._header_len, slot["work"].data_ptr(), slot["scratch"].data_ptr(),
                                         base + w._lengths_at, ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    ms = timed(files_deflate)
    lengths = slot["dev"][w._lengths_at:w._lengths_at + 24].cpu().numpy().view("uint64").tolist()
    row("gsr_frame_files_deflate (pack + previews + 7 kernels for the three compressed PNGs: 9 launches)", ms,
        (16 + 4 + 12) * W * H + 10 * W * H * 3 + sum(lengths) + 4 * W * H,
        f"32 B/pixel of float images in, 10 B/pixel of 8-bit images written, read, filtered and read again, the files out (PNGs {lengths} bytes of "
        f"{[frame_io.png_size(W, H, c) for c in (4, 3, 3)]} stored)")
    w.close()
big_c = torch.randint(0, 256, (2 * H, 2 * W, 4), dtype=torch.uint8, device=dev, generator=g)
big_d = torch.rand(2 * H, 2 * W, device=dev, generator=g) * 5
dst_c, dst_d = torch.empty(H, W, 4, dtype=torch.uint8, device=dev), torch.empty(H, W, device=dev)
row("gsr_resize_rgba8_bilinear 2x -> 1x (two passes)", timed(lambda: compositor.resize_rgba8(big_c, (W, H), out=dst_c)),
    16 * W * H + 2 * 8 * W * H + 4 * W * H, "16 B/out-pixel in, the [2H, W] intermediate written and read, 4 B out")
row("gsr_resize_f32_nearest 2x -> 1x", timed(lambda: compositor.resize_depth(big_d, (W, H), out=dst_d)), 8 * W * H, "4 B read + 4 B written per output pixel")
