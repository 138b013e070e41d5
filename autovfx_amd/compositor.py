"""GPU compositor for AutoVFX's final blend (``blender/blend_all.py::blend_frames``).

``composite_frame`` takes the layers of one frame as GPU tensors -- RGBA8 ``[H,W,4]`` colour layers and fp32
``[H,W]`` depth maps -- and returns the composited RGBA8 frame, bit-identical to the reference's numpy arithmetic
(``blend_all.py:236-300,341-343``).  Layers that arrive at Blender's render resolution (the reference renders them at 2x
for anti-aliasing) are brought to the background's size first, as ``blend_all.py:217-234`` does with PIL --
``downsample_image``: ``Image.resize(new_size, BILINEAR)`` for the colour layers, ``NEAREST`` for the depth maps -- by
``resize_rgba8`` / ``resize_depth``, which reproduce Pillow's results bit for bit on the GPU (``gsr_resize_rgba8_bilinear``,
``gsr_resize_f32_nearest``).  File discovery and PNG / EXR decoding (``:124-205``) stay with the caller (host libraries).
``smoke_depth_fill`` is the one non-pointwise step (``:207-215``): where the smoke layer has alpha its depth
becomes the layer's 0.001-th percentile, computed here with numpy's "linear" rule.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib


def _percentile_linear(values: torch.Tensor, q: float) -> torch.Tensor:
    """``numpy.percentile(values, q)`` for a float32 array with the default "linear" rule.  numpy keeps the whole
    computation in the array's dtype (virtual index, its fractional part and the lerp), so this does too."""
    import numpy as np
    v = values.reshape(-1).to(torch.float32).sort().values
    n = v.numel()
    pos = (np.float32(q) / np.float32(100)) * np.float32(n - 1)
    lo = int(np.floor(pos))
    hi = min(lo + 1, n - 1)
    t = float(np.float32(pos - np.float32(lo)))
    a, b = v[lo], v[hi]
    d = b - a
    one_minus_t = float(np.float32(1) - np.float32(t))
    return b - d * one_minus_t if t >= 0.5 else a + d * t   # numpy's _lerp, fp32 throughout


def smoke_depth_fill(s_f_c: torch.Tensor, s_f_d: torch.Tensor) -> torch.Tensor:
    mask = (s_f_c[..., 3].to(torch.float32) / 255.0) > 0.0
    return torch.where(mask, _percentile_linear(s_f_d, 0.001), s_f_d)    # (no boolean-mask assignment: that one waits for the GPU)


def resize_rgba8(image: torch.Tensor, size_wh, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``np.array(Image.fromarray(image).resize(size_wh, resample=Image.BILINEAR))`` for an RGBA8 ``[H,W,4]`` GPU tensor
    (``downsample_image`` of blend_all.py:21-28): same bytes as Pillow."""
    if not (image.is_cuda and image.dtype == torch.uint8 and image.dim() == 3 and image.shape[2] == 4):
        raise RuntimeError("resize_rgba8 expects a uint8 GPU tensor [H,W,4]")
    W, H = int(size_wh[0]), int(size_wh[1])
    src = image.contiguous()
    Hs, Ws = int(src.shape[0]), int(src.shape[1])
    if out is None:
        out = torch.empty((H, W, 4), dtype=torch.uint8, device=src.device)
    tmp = torch.empty((Hs, W, 4), dtype=torch.uint8, device=src.device) if (Ws != W and Hs != H) else None
    with torch.cuda.device(src.device):
        rc = _lib.lib.gsr_resize_rgba8_bilinear(src.data_ptr(), Ws, Hs, out.data_ptr(), W, H, None if tmp is None else tmp.data_ptr(),
                                                ctypes.c_void_p(torch.cuda.current_stream(src.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gsr_resize_rgba8_bilinear failed ({rc}): {_lib.last_error()}")
    return out


def resize_depth(depth: torch.Tensor, size_wh, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``np.array(Image.fromarray(depth).resize(size_wh, Image.NEAREST))`` for an fp32 ``[H,W]`` GPU tensor: same values as Pillow."""
    if not (depth.is_cuda and depth.dtype == torch.float32 and depth.dim() == 2):
        raise RuntimeError("resize_depth expects a float32 GPU tensor [H,W]")
    W, H = int(size_wh[0]), int(size_wh[1])
    src = depth.contiguous()
    if out is None:
        out = torch.empty((H, W), dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        rc = _lib.lib.gsr_resize_f32_nearest(src.data_ptr(), int(src.shape[1]), int(src.shape[0]), out.data_ptr(), W, H,
                                             ctypes.c_void_p(torch.cuda.current_stream(src.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gsr_resize_f32_nearest failed ({rc}): {_lib.last_error()}")
    return out


def _layer(t: Optional[torch.Tensor], dtype, shape, device, name):
    """A layer at the frame's size: as it is; at another size (Blender's render resolution): resized as the reference does."""
    if t is None:
        return None
    if t.dtype != dtype or t.device != device or t.dim() != len(shape) or (len(shape) == 3 and t.shape[2] != 4):
        raise RuntimeError(f"{name}: expected {dtype} {shape} on {device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
    if tuple(t.shape) != shape:
        size_wh = (shape[1], shape[0])
        return resize_rgba8(t, size_wh) if dtype == torch.uint8 else resize_depth(t, size_wh)
    return t.contiguous()


def composite_frame(bg_c, o_c, o_d, s_c, s_d, o_s_c, o_gs_c=None, o_gs_d=None, s_f_c=None, s_f_d=None,
                    s_f_c_pre=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if not bg_c.is_cuda:
        raise RuntimeError("composite_frame: layers must live on a HIP device (there is no CPU fallback)")
    dev, (H, W) = bg_c.device, bg_c.shape[:2]
    rgba, depth = (H, W, 4), (H, W)
    L = [_layer(bg_c, torch.uint8, rgba, dev, "bg_c"), _layer(o_c, torch.uint8, rgba, dev, "o_c"),
         _layer(o_d, torch.float32, depth, dev, "o_d"), _layer(s_c, torch.uint8, rgba, dev, "s_c"),
         _layer(s_d, torch.float32, depth, dev, "s_d"), _layer(o_s_c, torch.uint8, rgba, dev, "o_s_c"),
         _layer(o_gs_c, torch.uint8, rgba, dev, "o_gs_c"), _layer(o_gs_d, torch.float32, depth, dev, "o_gs_d"),
         _layer(s_f_c, torch.uint8, rgba, dev, "s_f_c"), _layer(s_f_d, torch.float32, depth, dev, "s_f_d"),
         _layer(s_f_c_pre, torch.uint8, rgba, dev, "s_f_c_pre")]
    if out is None:
        out = torch.empty(rgba, dtype=torch.uint8, device=dev)
    ptr = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(dev):
        rc = _lib.lib.gsr_composite(int(W), int(H), *[ptr(t) for t in L], out.data_ptr(),
                                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"gsr_composite failed ({rc}): {_lib.last_error()}")
    return out


# ---- blend_all.blend_frames, the function scene_representation.py:232 calls ---------------------------------------------------

def load_rgb(path):
    """``blend_all.load_rgb`` (:56-60): RGBA8 ``[H,W,4]`` or None when the layer does not exist.  PNG decoding is a host library's job
    (Pillow, as in the reference)."""
    import os
    if not os.path.exists(path):
        return None
    import numpy as np
    from PIL import Image
    return np.array(Image.open(path).convert("RGBA"))


def load_depth_exr(path):
    """``blend_all.load_depth_exr`` (:70-75): Blender's OpenEXR depth pass as ``cv2.imread(path, ANYCOLOR | ANYDEPTH)[:, :, 0]`` gives
    it -- float32 ``[H,W]`` of the file's B channel.  With OpenCV installed (``OPENCV_IO_ENABLE_OPENEXR=1``) that very call is made,
    as in the reference; without it ``autovfx_amd.exr`` reads the file (scanline files, half / float, NONE / RLE / ZIPS / ZIP: what
    Blender's File Output node writes by default)."""
    import os
    if not os.path.exists(path):
        return None
    os.environ.setdefault("OPENCV_IO_ENABLE_OPENEXR", "1")
    try:
        import cv2
    except ImportError:
        cv2 = None
    if cv2 is not None and hasattr(cv2, "imread"):
        d = cv2.imread(path, cv2.IMREAD_ANYCOLOR | cv2.IMREAD_ANYDEPTH)
        return d[:, :, 0]
    from .exr import load_depth_exr as read
    return read(path)


_LAYERS_RGB = ("rgb_obj", "rgb_shadow", "rgb_all", "rgb_obj_3dgs", "rgb_smoke_fire", "rgb_smoke_fire_pre")
_LAYERS_DEPTH = ("depth_obj", "depth_shadow", "depth_obj_3dgs", "depth_smoke_fire")


def _load_frame_layers(cache, bg_path, i):
    """Everything ``blend_frames`` reads for frame ``i`` (blend_all.py:185-205), decoded on the host with the reference's libraries: a
    dict of numpy arrays / None (the reference-shaped loader: tests and bench.py compare against it)."""
    import os
    out = {"bg": load_rgb(bg_path)}
    for kind in _LAYERS_RGB:
        out[kind] = load_rgb(os.path.join(cache, kind, "{:0>3d}.png".format(i + 1)))
    for kind in _LAYERS_DEPTH:
        out[kind] = load_depth_exr(os.path.join(cache, kind, "{:0>3d}".format(i + 1), "Image{:0>4d}.exr".format(i + 1)))
    return out


_worker_local = None


def _thread_stream(dev):
    """One side stream per pool thread and device: whatever the thread queues for its frame is ordered among itself and runs beside
    the other threads' frames."""
    import threading
    global _worker_local
    if _worker_local is None:
        _worker_local = threading.local()
    streams = _worker_local.__dict__.setdefault("streams", {})
    stream = streams.get(dev)
    if stream is None:
        stream = streams[dev] = torch.cuda.Stream(device=dev)
    return stream


def _load_frame_layers_to_gpu(cache, bg_path, i, dev, staging, error_flag=None):
    """The same layers as GPU tensors (``uint8[H, W, 4]``; depth in the file's precision), queued on the CURRENT stream: the host
    inflates the files' zlib streams into ``staging`` (page-locked) and copies from there, kernels undo the PNG filters / the EXR
    predictor (``autovfx_amd.layer_io``).  ``staging`` may be reset once the stream has been synchronised.  With ``error_flag`` the
    EXR passes' zlib streams are inflated on the GPU as well (a refused stream sets the flag)."""
    import os
    from . import layer_io
    pngs = [bg_path] + [os.path.join(cache, kind, "{:0>3d}.png".format(i + 1)) for kind in _LAYERS_RGB]
    out = dict(zip(("bg",) + _LAYERS_RGB, layer_io.load_rgba_many(pngs, dev, staging)))       # (one launch for the frame's PNGs)
    exrs = [os.path.join(cache, kind, "{:0>3d}".format(i + 1), "Image{:0>4d}.exr".format(i + 1)) for kind in _LAYERS_DEPTH]
    out.update(zip(_LAYERS_DEPTH, layer_io.load_depth_many(exrs, dev, staging, error_flag)))    # (one inflate launch for the four passes)
    return out


def _composite_layers(L):
    """The loaded layers of a frame -> the blended RGBA8 frame on the GPU (blend_all.py:207-337), queued on the current stream."""
    bg_c = L["bg"]
    o_c, o_d = L["rgb_obj"], L["depth_obj"]
    s_c, s_d = L["rgb_shadow"], L["depth_shadow"]
    o_s_c = L["rgb_all"]
    o_gs_c, o_gs_d = L["rgb_obj_3dgs"], L["depth_obj_3dgs"]
    s_f_c, s_f_d = L["rgb_smoke_fire"], L["depth_smoke_fire"]
    s_f_c_pre = L["rgb_smoke_fire_pre"]
    if o_gs_c is None:
        o_gs_d = None
    if s_f_c is not None:
        s_f_d = smoke_depth_fill(s_f_c, s_f_d.to(torch.float32))       # on the full-size layers, before the resizes (:207-215)
    else:
        s_f_d = s_f_c_pre = None
    f32 = lambda t: None if t is None else t.to(torch.float32)
    return composite_frame(bg_c, o_c, f32(o_d), s_c, f32(s_d), o_s_c, o_gs_c, f32(o_gs_d), s_f_c, f32(s_f_d), s_f_c_pre)


def _blend_one_frame(cache, bg_path, i, dev, out_path, want_frame, stats=None, exr_on_gpu=None):
    """Frame ``i`` of ``blend_frames`` from its files to its file (blend_all.py:185-337), start to finish on the calling thread and its
    stream (the pool's workers split this in two -- ``_FrameWorker`` -- and come back here, with zlib on the host, for a frame whose
    depth passes the GPU's decoder refused)."""
    from . import frame_io, layer_io
    if exr_on_gpu is None:
        exr_on_gpu = layer_io.exr_inflate_on_gpu()
    staging = layer_io.Staging()
    with torch.cuda.stream(_thread_stream(dev)):
        error_flag = torch.zeros(1, dtype=torch.int32, device=dev) if exr_on_gpu else None
        frame = _composite_layers(_load_frame_layers_to_gpu(cache, bg_path, i, dev, staging, error_flag))
        png = frame_io.encode_png_gpu_deflate(frame) if frame_io.deflate_default() else frame_io.encode_png_gpu(frame)
        data = png.cpu().numpy()                   # (waits for this thread's stream)
        if error_flag is not None and int(error_flag.cpu()) != 0:
            return _blend_one_frame(cache, bg_path, i, dev, out_path, want_frame, stats, exr_on_gpu=False)
        host_frame = frame.cpu().numpy() if want_frame else None
    with open(out_path, "wb") as f:
        f.write(data)
    return host_frame


class _FrameWorker:
    """One pool thread of ``blend_frames``: takes whole frames off a shared counter and keeps TWO in flight -- while the GPU undoes the
    predictors of frame i, composites and encodes it (5 - 15 ms behind the other threads' kernels), the thread already reads and
    inflates the files of its next frame; only then does it wait for frame i, and writes its file.  Two staging arenas and two sets
    of page-locked result buffers alternate."""

    def __init__(self, job):
        self.job = job
        self.slots = [{"staging": None, "png": None, "len": None, "flag": None, "frame": None} for _ in range(2)]

    def _pinned(self, slot, key, numel, dtype):
        t = slot[key]
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = slot[key] = torch.empty(numel, dtype=dtype, pin_memory=True)
        return t

    def begin(self, i, slot):
        import time
        from . import frame_io, layer_io
        job, dev = self.job, self.job["dev"]
        t0 = time.perf_counter()
        if slot["staging"] is None:
            slot["staging"] = layer_io.Staging()
        slot["staging"].reset()                     # (this slot's previous frame has been waited for)
        stream = _thread_stream(dev)
        with torch.cuda.stream(stream):
            flag = torch.zeros(1, dtype=torch.int32, device=dev) if job["exr_on_gpu"] else None
            L = _load_frame_layers_to_gpu(job["cache"], job["bg_rgb"][i], i, dev, slot["staging"], flag)
            t1 = time.perf_counter()
            frame = _composite_layers(L)
            if frame_io.deflate_default():
                out, length = frame_io.encode_png_gpu_deflate_queued(frame)
                self._pinned(slot, "len", 1, torch.int64).copy_(length, non_blocking=True)
                n = None
            else:
                out = frame_io.encode_png_gpu(frame)
                n = out.numel()
            self._pinned(slot, "png", out.numel(), torch.uint8)[:out.numel()].copy_(out, non_blocking=True)
            if flag is not None:
                self._pinned(slot, "flag", 1, torch.int32).copy_(flag, non_blocking=True)
            if job["want_frames"]:
                self._pinned(slot, "frame", frame.numel(), torch.uint8)[:frame.numel()].copy_(frame.reshape(-1), non_blocking=True)
            done = torch.cuda.Event()
            done.record(stream)
        t2 = time.perf_counter()
        return {"i": i, "slot": slot, "done": done, "n": n, "checked": flag is not None, "shape": tuple(frame.shape), "t": (t1 - t0, t2 - t1)}

    def finish(self, fr):
        import time
        job, slot, i = self.job, fr["slot"], fr["i"]
        t0 = time.perf_counter()
        fr["done"].synchronize()
        t1 = time.perf_counter()
        if fr["checked"] and int(slot["flag"][0]) != 0:
            # the GPU's decoder refused a depth pass's zlib stream: once more with zlib on the host, which reads it or says why not
            host_frame = _blend_one_frame(job["cache"], job["bg_rgb"][i], i, job["dev"], job["paths"][i], job["want_frames"], None, exr_on_gpu=False)
        else:
            n = fr["n"] if fr["n"] is not None else int(slot["len"][0])
            with open(job["paths"][i], "wb") as f:
                f.write(memoryview(slot["png"].numpy())[:n])
            host_frame = slot["frame"][:int(torch.Size(fr["shape"]).numel())].numpy().reshape(fr["shape"]).copy() if job["want_frames"] else None
        if job["want_frames"]:
            job["host_frames"][i] = host_frame
        if job["stats"] is not None:
            job["stats"].append(fr["t"] + (t1 - t0, time.perf_counter() - t1))

    def run(self):
        job, prev, k = self.job, None, 0
        try:
            while job["error"] is None:
                with job["lock"]:
                    i = job["next"]
                    job["next"] += 1
                if i >= job["n_frame"]:
                    break
                cur = self.begin(i, self.slots[k & 1])
                k += 1
                if prev is not None:
                    self.finish(prev)
                prev = cur
            if prev is not None and job["error"] is None:
                self.finish(prev)
        except BaseException as e:                  # the first failure stops the pool and is raised by blend_frames
            with job["lock"]:
                if job["error"] is None:
                    job["error"] = e


LAST_BLEND_STATS: dict = {}     # with AUTOVFX_AMD_BLEND_STATS=1: thread-seconds of the last blend_frames call by stage, summed over the pool


def decode_threads() -> int:
    """Threads of ``blend_frames``' pool: ``AUTOVFX_AMD_BLEND_DECODERS``, or the cores this process may run on, at most 16 (past that
    the interpreter lock around the threads' Python stretches eats the gain)."""
    import os
    n = int(os.environ.get("AUTOVFX_AMD_BLEND_DECODERS", "0"))
    if n <= 0:
        n = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 4)
    return max(1, n)


def blend_frames(blend_results_dir, input_config_path=None, device=None, write_video=True):
    """Drop-in for ``blender/blend_all.py::blend_frames`` (:95-346; called at ``scene_representation.py:232``): same arguments, same
    files found the same way (the 3DGS frames under ``<root>/images/*.png``, Blender's layers under
    ``<blender_cache_dir>/<output_dir_name>/{rgb,depth}_*``, the frame count from ``rgb_all/*.png``), same outputs
    (``<blend_results_dir>/frames/%04d.png`` and, when ``imageio`` / ``skimage`` are importable, ``blended.mp4``).  What happens
    between loading and saving runs on the GPU: the smoke-depth fill, PIL's resizes of every Blender layer to the frame's size
    (``resize_rgba8`` / ``resize_depth``), the per-pixel composite (``gsr_composite``) and the PNG encoding of the result
    (``gsr_png_encode_deflate``: compressed files holding the same pixels as ``Image.fromarray(frame).save``).  Of reading the layers
    the host keeps the container parsing and the PNGs' ``inflate``; the EXR blocks' zlib streams, the PNG filters and the EXR predictor
    are undone on the GPU (``autovfx_amd.layer_io``); files those kernels do not cover go to Pillow / OpenCV / ``autovfx_amd.exr``.
    Returns the list of frame paths."""
    import glob
    import json
    import os
    import numpy as np
    from . import frame_io
    root_dir = os.path.dirname(os.path.normpath(os.path.dirname(os.path.normpath(blend_results_dir))))   # up two levels (:97)
    assert input_config_path is not None, "input_config is required for blending frames"
    with open(input_config_path, "r") as f:
        input_config = json.load(f)
    cache = os.path.join(input_config["blender_cache_dir"], input_config["output_dir_name"])
    bg_rgb = sorted(glob.glob(os.path.join(root_dir, "images", "*.png")))
    n_frame = len(glob.glob(os.path.join(cache, "rgb_all", "*.png")))     # "ensure an output video even if Blender crashes" (:126)
    out_img_dir = os.path.join(blend_results_dir, "frames")
    os.makedirs(out_img_dir, exist_ok=True)
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    video = None
    if write_video:       # the frames are kept on the host only if the video can be written at all (400 frames of 960x540 are 0.8 GB)
        try:
            import imageio.v2 as imageio
            import skimage.transform
            video = (imageio, skimage.transform)
        except ImportError:
            print("[autovfx_amd] blended.mp4 will not be written: imageio / skimage are not installed (the frames go to " + out_img_dir + ")")
    # Reading ~11 PNG / EXR layers per frame at Blender's resolution dwarfs the 0.13 ms of resizing, compositing and encoding a frame,
    # and frames do not depend on each other: each thread of a pool takes whole frames -- inflates and uploads the files, queues the
    # kernels that inflate the EXR blocks, undo the image predictors (layer_io), resize, composite and encode on ITS stream, copies
    # the file out and writes it -- two frames in flight per thread (_FrameWorker).  This thread only waits.
    import threading
    from . import layer_io
    workers = max(1, min(decode_threads(), n_frame))
    paths = [os.path.join(out_img_dir, "{:0>4d}.png".format(i)) for i in range(n_frame)]
    job = {"cache": cache, "bg_rgb": bg_rgb, "dev": dev, "paths": paths, "n_frame": n_frame, "next": 0, "lock": threading.Lock(), "error": None,
           "want_frames": video is not None, "host_frames": {}, "exr_on_gpu": layer_io.exr_inflate_on_gpu(),
           "stats": [] if os.environ.get("AUTOVFX_AMD_BLEND_STATS") == "1" else None}
    threads = [threading.Thread(target=_FrameWorker(job).run, name=f"blend-frame-{k}") for k in range(workers)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.empty_cache()       # (the workers' streams are gone: what the allocator cached for them would never be handed out again)
    if hasattr(torch._C, "_host_emptyCache"):
        torch._C._host_emptyCache()                 # ... and their page-locked staging (~100 MB per worker)
    if job["error"] is not None:
        raise job["error"]
    host_frames = [job["host_frames"][i] for i in range(n_frame)] if video is not None else []
    stats = job["stats"]
    if stats is not None:
        LAST_BLEND_STATS.clear()
        LAST_BLEND_STATS.update({"frames": n_frame, "threads": workers, "read_and_upload_s": sum(x[0] for x in stats),
                                 "queue_kernels_s": sum(x[1] for x in stats), "wait_for_gpu_s": sum(x[2] for x in stats),
                                 "write_file_s": sum(x[3] for x in stats)})
    if video is not None and host_frames:   # generate_video_from_frames (:31-54), with the reference's own host libraries
        imageio, transform = video
        h, w = host_frames[0].shape[:2]
        new_h, new_w = h - h % 2, w - w % 2
        series = [(transform.resize(fr, (new_h, new_w)) * 255.0).astype(np.uint8) for fr in host_frames]
        imageio.mimsave(os.path.join(blend_results_dir, "blended.mp4"), series, fps=15, macro_block_size=1)
    return paths
