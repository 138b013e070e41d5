#!/usr/bin/env python
"""One Blender layer from file to GPU memory (autovfx_amd.layer_io): where its time goes, beside the host decoders the reference uses.
A 1920x1080 RGBA PNG as Pillow writes it and a four-channel half-float ZIP OpenEXR depth pass, as in bench.py's synthetic Blender tree.
Prints one JSON line."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from autovfx_amd import compositor, exr, layer_io  # noqa: E402


def _timed(fn, n):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from PIL import Image
    dev = torch.device("cuda", 0)
    W, H = 1920, 1080
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
    img = np.empty((H, W, 4), np.uint8)
    for c in range(3):
        img[..., c] = np.clip((0.7 - 0.1 * c) * (0.6 + 0.4 * np.sin(xx * 0.01 + c)) * 255, 0, 255).astype(np.uint8)
    img[..., 3] = (np.clip(1.5 - np.hypot(xx - 0.45 * W, yy - 0.5 * H) / (0.2 * H), 0, 1) * 255).astype(np.uint8)
    z = (3.0 + 2.0 * np.clip(1.5 - np.hypot(xx - 0.5 * W, yy - 0.5 * H) / (0.5 * H), 0, 1) + 0.2 * np.sin(yy * 0.02)).astype(np.float32)
    d = tempfile.mkdtemp(prefix="gsr_layer_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    png, exrp = os.path.join(d, "layer.png"), os.path.join(d, "Image0001.exr")
    Image.fromarray(img).save(png, compress_level=1)
    exr.write_exr(exrp, {"R": z, "G": z, "B": z, "A": np.ones_like(z)}, half=True, level=4)      # (zlib level 4: OpenEXR 3.1.3+)
    staging = layer_io.Staging(64 << 20)
    out = {"png": {"file_bytes": os.path.getsize(png)}, "exr": {"file_bytes": os.path.getsize(exrp)}}

    def gpu_ms(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            staging.reset()
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    # host side alone: parse + inflate into page-locked memory
    buf = open(png, "rb").read()
    w, h, c, stream = layer_io.png_chunks(buf)
    host = staging.take(h * (1 + w * c))
    out["png"]["host_parse_inflate_python_ms"] = round(_timed(lambda: (layer_io.png_chunks(buf), layer_io.inflate_into(host, stream)), 20), 3)

    def native_png():
        staging.reset()
        return layer_io.read_png_scanlines(buf, staging)
    out["png"]["host_parse_inflate_ms"] = round(_timed(native_png, 20), 3)           # the product path: two native calls
    host = native_png()[0]
    staged = layer_io._upload(host, dev)
    rgba = torch.empty((h, w, 4), dtype=torch.uint8, device=dev)
    scratch = torch.empty(layer_io._lib.lib.gsr_png_unfilter_scratch(w, h), dtype=torch.uint8, device=dev)
    import ctypes
    stream_ptr = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    out["png"]["upload_ms"] = round(gpu_ms(lambda: layer_io._upload(host, dev)), 4)
    out["png"]["unfilter_kernels_ms"] = round(gpu_ms(lambda: layer_io._lib.lib.gsr_png_unfilter(staged.data_ptr(), w, h, c, rgba.data_ptr(), scratch.data_ptr(), stream_ptr)), 4)
    out["png"]["load_rgba_blocking_ms"] = round(_timed(lambda: layer_io.load_rgba(png, dev), 10), 3)
    out["png"]["pillow_load_rgb_ms"] = round(_timed(lambda: compositor.load_rgb(png), 5), 3)
    ok = bool(np.array_equal(layer_io.load_rgba(png, dev).cpu().numpy(), compositor.load_rgb(png)))
    # seven layers (a frame's PNGs) in one batch launch: a workgroup = a compute unit per layer, side by side
    outs = [torch.empty_like(rgba) for _ in range(7)]
    scr = [torch.empty_like(scratch) for _ in range(7)]
    table = (layer_io._lib.PngUnfilterJob * 7)(*[layer_io._lib.PngUnfilterJob(staged.data_ptr(), w, h, c, o.data_ptr(), sc.data_ptr()) for o, sc in zip(outs, scr)])
    out["png"]["unfilter_batch_of_7_ms"] = round(gpu_ms(lambda: layer_io._lib.lib.gsr_png_unfilter_batch(7, ctypes.byref(table), stream_ptr)), 4)
    out["png"]["same_as_pillow"] = ok

    ebuf = open(exrp, "rb").read()
    L, _pick, _name, pieces = layer_io._exr_plan(ebuf)
    total = L["height"] * L["bytes_per_line"]
    ehost = staging.take(total)

    def inflate_all():
        at = 0
        for data, expected in pieces:
            layer_io.inflate_into(ehost[at:at + expected], data)
            at += expected
    out["exr"]["host_parse_inflate_python_ms"] = round(_timed(lambda: (layer_io._exr_plan(ebuf), inflate_all()), 10), 3)

    def native_exr():
        staging.reset()
        return layer_io.read_exr_blocks(ebuf, staging)
    out["exr"]["host_parse_inflate_ms"] = round(_timed(native_exr, 10), 3)           # the product path: two native calls
    ehost = native_exr()[0]
    estaged = layer_io._upload(ehost, dev)
    plane = torch.empty((L["height"], L["channel_bytes"]), dtype=torch.uint8, device=dev)
    out["exr"]["upload_ms"] = round(gpu_ms(lambda: layer_io._upload(ehost, dev)), 4)
    out["exr"]["unpack_kernel_ms"] = round(gpu_ms(lambda: layer_io._lib.lib.gsr_exr_unpack_channel(
        estaged.data_ptr(), L["height"], L["bytes_per_line"], L["lines_per_block"], L["channel_at"], L["channel_bytes"], plane.data_ptr(), stream_ptr)), 4)
    out["exr"]["load_depth_blocking_ms"] = round(_timed(lambda: layer_io.load_depth(exrp, dev), 10), 3)
    # the same file with its 68 zlib streams inflated on the GPU (a single-wave workgroup each): host side = copying the streams together
    flag = torch.zeros(1, dtype=torch.int32, device=dev)

    def pack_only():
        staging.reset()
        info = layer_io._lib.ExrFileInfo()
        layer_io._lib.lib.gsr_exr_file_probe(ebuf, len(ebuf), None, ctypes.byref(info))
        room = len(ebuf) + 4 * info.n_blocks + 4
        h = staging.take(16 * info.n_blocks + room)
        n = ctypes.c_size_t(0)
        layer_io._lib.lib.gsr_exr_file_pack(ebuf, len(ebuf), None, ctypes.c_void_p(h.data_ptr() + 16 * info.n_blocks), room, ctypes.c_void_p(h.data_ptr()), ctypes.byref(n))
        return h, info, n.value
    out["exr"]["gpu_inflate_host_pack_ms"] = round(_timed(pack_only, 20), 3)
    h, info, n_packed = pack_only()
    d_in = layer_io._upload(h[:16 * info.n_blocks + n_packed + 4], dev)
    d_blocks = torch.empty(info.blocks_bytes, dtype=torch.uint8, device=dev)
    d_status = torch.empty(info.n_blocks, dtype=torch.int32, device=dev)
    out["exr"]["gpu_inflate_kernel_ms"] = round(gpu_ms(lambda: layer_io._lib.lib.gsr_inflate_zlib_blocks(
        d_in.data_ptr() + 16 * info.n_blocks, d_blocks.data_ptr(), d_in.data_ptr(), info.n_blocks, d_status.data_ptr(), flag.data_ptr(), stream_ptr)), 4)
    out["exr"]["gpu_inflate_streams"] = int(info.n_blocks)
    out["exr"]["load_depth_gpu_inflate_blocking_ms"] = round(_timed(lambda: layer_io.load_depth(exrp, dev, None, flag), 10), 3)
    out["exr"]["gpu_inflate_same_as_host_reader"] = bool(int(flag.cpu()) == 0 and np.array_equal(
        layer_io.load_depth(exrp, dev, None, flag).to(torch.float32).cpu().numpy(), exr.load_depth_exr(exrp)))
    out["exr"]["host_reader_ms"] = round(_timed(lambda: exr.load_depth_exr(exrp), 5), 3)
    out["exr"]["same_as_host_reader"] = bool(np.array_equal(layer_io.load_depth(exrp, dev).to(torch.float32).cpu().numpy(), exr.load_depth_exr(exrp)))
    print(json.dumps(out))
    import shutil
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
