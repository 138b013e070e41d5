"""Frame-parallel rendering of a camera trajectory: one process per GPU, frames dealt round-robin,
one gather of the finished frames to rank 0 at the end (RCCL over xGMI when the backend is "nccl").

The reference renders a trajectory serially on one GPU (``scene_representation.py:355-356``: a
``tqdm`` loop over ``camera_views``); frames are independent when the Gaussian set is static, so
the only multi-GPU strategy that fits is data parallelism over frames.  Every
rank holds a full replica of the Gaussians (3 M Gaussians = 0.7 GB; HBM is 288 GB), rank ``r`` of
``N`` renders frames ``r, r+N, r+2N, ...`` (round-robin balances an orbit's slowly varying load
better than contiguous blocks) and nothing is exchanged until the final gather.  No collective sits
on the per-frame path.

Frame payload: RGBA8, quantised like ``torchvision.utils.save_image`` does for the PNGs the
reference writes (``scene_representation.py:427``), i.e. ``clamp(x*255+0.5, 0, 255)``
truncated to uint8; optionally the fp32 depth map (what ``depth/*.npy`` holds).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.distributed as dist

from .cameras import Camera
from .scenes import GaussianCloud


def shard_frames(num_frames: int, rank: int, world_size: int) -> List[int]:
    """Frame indices owned by ``rank``: round-robin."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    return list(range(rank, num_frames, world_size))


def local_device() -> torch.device:
    """The GPU this process drives: one process per GPU, ``cuda:LOCAL_RANK`` as ``torch.distributed.run`` hands it out
    (``LOCAL_RANK``, not ``RANK``: on a second node rank 11 is device 3)."""
    import os
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))


def rank_world() -> tuple:
    """``(RANK, WORLD_SIZE)`` from the launcher's environment (``(0, 1)`` without one)."""
    import os
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def pack_rgba8(color: torch.Tensor, alpha: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[3,H,W] + [1,H,W] float -> [4,H,W] uint8 with save_image's rounding.  GPU tensors go through
    the library's fused kernel (one launch, 20 B per pixel); anything else through plain torch ops."""
    if color.is_cuda and color.dtype == torch.float32 and alpha.dtype == torch.float32:
        import ctypes
        from . import _lib
        H, W = int(color.shape[-2]), int(color.shape[-1])
        if out is None:
            out = torch.empty((4, H, W), dtype=torch.uint8, device=color.device)
        if not (out.is_contiguous() and out.dtype == torch.uint8 and out.device == color.device):
            raise RuntimeError("pack_rgba8: out must be a contiguous uint8 tensor on the colour tensor's device")
        c, a = color.contiguous(), alpha.contiguous()
        with torch.cuda.device(color.device):
            rc = _lib.lib.gsr_pack_rgba8(c.data_ptr(), a.data_ptr(), out.data_ptr(), W, H,
                                         ctypes.c_void_p(torch.cuda.current_stream(color.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(f"gsr_pack_rgba8 failed ({rc}): {_lib.last_error()}")
        return out
    rgba = torch.cat((color, alpha), dim=0)
    q = rgba.mul(255.0).add_(0.5).clamp_(0.0, 255.0).to(torch.uint8)
    if out is not None:
        out.copy_(q)
        return out
    return q


def settings_for_camera(cam: Camera, bg: torch.Tensor, sh_degree: int, scale_modifier: float = 1.0,
                        debug: bool = False):
    """The ``GaussianRasterizationSettings`` that ``gaussian_renderer.render`` builds
    (``gaussian_renderer/__init__.py:98-114``)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=cam.tanfovx,
        tanfovy=cam.tanfovy, bg=bg, scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform,
        projmatrix=cam.full_proj_transform, sh_degree=sh_degree, campos=cam.camera_center, prefiltered=False,
        debug=debug)


def rasterize(cloud: GaussianCloud, cam: Camera, bg: torch.Tensor):
    """One forward call through the drop-in API; ``cloud``/``cam``/``bg`` already on the GPU."""
    from diff_gaussian_rasterization import GaussianRasterizer
    rast = GaussianRasterizer(raster_settings=settings_for_camera(cam, bg, cloud.sh_degree))
    if cloud.colors_precomp is not None:
        return rast(means3D=cloud.means3D, means2D=None, opacities=cloud.opacities, colors_precomp=cloud.colors_precomp,
                    scales=cloud.scales, rotations=cloud.rotations)
    return rast(means3D=cloud.means3D, means2D=None, opacities=cloud.opacities, shs=cloud.shs, scales=cloud.scales,
                rotations=cloud.rotations)


class PendingFrame:
    """A frame whose first half is queued (``rasterize_begin``); ``finish()`` queues the rest and returns
    ``(color, depth, alpha, radii)`` like ``rasterize``."""

    def __init__(self, pending, keep):
        self._pending, self._keep = pending, keep

    def ready(self) -> bool:
        return self._pending.ready()

    def finish(self):
        _n, color, depth, alpha, radii = self._pending.finish()[:5]
        self._keep = None
        return color, depth, alpha, radii


def rasterize_begin(cloud: GaussianCloud, cam: Camera, bg: torch.Tensor) -> PendingFrame:
    """``rasterize`` in two halves, for inference only (no autograd graph is recorded): projection and depth sort
    are queued on the current stream and the call returns without waiting for the GPU, so one host thread can hold
    a frame in flight on each of several streams.  Arguments reach the library in the order
    ``_RasterizeGaussians.forward`` passes them (reference ``__init__.py:62-82``)."""
    from diff_gaussian_rasterization import _C
    s = settings_for_camera(cam, bg, cloud.sh_degree)
    absent = torch.empty(0, dtype=torch.float32, device=cloud.means3D.device)
    shs = absent if cloud.colors_precomp is not None else cloud.shs
    colors = cloud.colors_precomp if cloud.colors_precomp is not None else absent
    with torch.no_grad():
        pending = _C.rasterize_gaussians_begin(
            s.bg, cloud.means3D, colors, cloud.opacities, cloud.scales, cloud.rotations, s.scale_modifier, absent,
            s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy, s.image_height, s.image_width, shs, s.sh_degree,
            s.campos, s.prefiltered, s.debug, inference=True)
    return PendingFrame(pending, (s, absent))


_SIDE_STREAMS: Dict[tuple, List["torch.cuda.Stream"]] = {}


def side_streams(device, count: int) -> List["torch.cuda.Stream"]:
    """The HIP streams frames are rendered on, created once per device and reused by every call: torch's caching
    allocator keeps one pool per stream, so fresh streams would pay for fresh device allocations of every frame's
    scratch (hundreds of MB at C3) each time a shard is rendered."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    have = _SIDE_STREAMS.setdefault(key, [])
    while len(have) < count:
        have.append(torch.cuda.Stream(device=device))
    return have[:count]


# Frames in flight per GPU (one HIP stream each, all fed by one host thread).  The HIP runtime spreads a process's streams
# over 4 hardware queues, so at most four frames run truly side by side; more streams than that let the host queue
# further ahead of the one read-back every call has.  Measured on MI355X, same box, C3: 3 / 7 / 11 / 15 streams =
# 1451 / 1475 / 1494 / 1487 frames/s, multiples of four are the worst (4: -8 %, 8: -3 %), and raising the queue count
# (GPU_MAX_HW_QUEUES=8) loses 6 - 9 %: more concurrent blends only fight over the vector ALUs.
DEFAULT_STREAMS = 11

RenderFn = Callable[[GaussianCloud, Camera, torch.Tensor], Sequence[torch.Tensor]]
# the split form of a RenderFn: returns an object whose ``finish()`` returns what the RenderFn returns
BeginFn = Callable[[GaussianCloud, Camera, torch.Tensor], object]


def render_shard(cloud: GaussianCloud, cameras: Sequence[Camera], frame_ids: Sequence[int], bg: torch.Tensor,
                 keep_depth: bool = False, render_fn: RenderFn = rasterize, streams: int = DEFAULT_STREAMS,
                 driver: str = "auto", begin_fn: Optional[BeginFn] = None,
                 chunk_ends: Sequence[int] = (), on_chunk: Optional[Callable[[int, torch.Tensor], None]] = None,
                 pad_to: int = 0) -> Dict[str, torch.Tensor]:
    """Render this rank's frames; returns stacked ``rgba8 [n,4,H,W]`` (and ``depth [n,H,W]``).

    ``on_chunk(end, rgba8[:end])`` is called -- in the caller's stream context, ordered after the frames it covers --
    as soon as frame ``end - 1`` is queued, for every ``end`` in ``chunk_ends``, while later frames keep rendering
    (``render_and_gather`` starts a piece's transfer from it).  Not available with ``driver="threads"``.

    ``pad_to`` > n: the stacks get ``pad_to`` rows, the rows past this rank's n frames zero (round-robin shards differ
    by at most one frame; a collective wants equal pieces), and chunk ends past n are reported once the last real
    frame is queued.

    ``streams > 1`` (GPU only) renders frame ``slot`` on HIP stream ``slot % streams``.  Frames are independent, and
    a frame's blend (VALU-bound) overlaps well with other frames' projection and sorts (HBM / latency-bound): several
    streams (default ``DEFAULT_STREAMS``) raise throughput by 40 % at C3 on MI355X without touching per-frame results.
    Two ways of feeding the streams:

    * ``driver="pipelined"``: ONE host thread.  Each frame's call is split where the host needs the pair count
      (``rasterize_begin`` / ``finish``): the thread queues the first half of frame ``i + streams - 1`` before it
      waits for the counters of frame ``i``, so the GPU always has the other streams' work queued and no second
      thread, lock or GIL hand-over sits on the frame path.  Needs the call in split form: the default ``render_fn``
      has one (``rasterize_begin``); for another ``render_fn`` pass its ``begin_fn`` (e.g. one built on
      ``renderer.render_begin``).
    * ``driver="threads"``: one host thread per stream, each making ordinary blocking calls (any ``render_fn``).

    ``"auto"`` picks ``pipelined`` when a split form is available.  The call returns after all streams drained.
    """
    device = cloud.means3D.device
    n = len(frame_ids)
    rows = max(n, int(pad_to))
    H, W = (cameras[0].image_height, cameras[0].image_width) if len(cameras) else (0, 0)
    rgba = torch.empty((rows, 4, H, W), dtype=torch.uint8, device=device)
    depth = torch.empty((rows, H, W), dtype=torch.float32, device=device) if keep_depth else None
    if rows > n:
        rgba[n:].zero_()
        if keep_depth:
            depth[n:].zero_()
    if driver not in ("auto", "pipelined", "threads"):
        raise ValueError(f"unknown driver {driver!r}")
    if begin_fn is None and render_fn is rasterize:
        begin_fn = rasterize_begin
    elif begin_fn is not None and render_fn is rasterize:   # only the split form was given: the blocking form is begin + finish
        split = begin_fn
        render_fn = lambda c, cam, b: split(c, cam, b).finish()
    if driver == "pipelined" and begin_fn is None:
        raise ValueError("the pipelined driver needs begin_fn, the split form of render_fn")
    if driver == "auto":
        driver = "pipelined" if begin_fn is not None else "threads"

    def keep(slot, color, d, alpha):
        pack_rgba8(color, alpha, out=rgba[slot])
        if keep_depth:
            depth[slot].copy_(d[0])

    chunk_ends = set(int(e) for e in chunk_ends) if on_chunk is not None else set()
    late_ends = sorted(e for e in chunk_ends if e > n)   # pieces that end in the padding rows

    def render_slots(slots, report=False):
        with torch.no_grad():
            for slot in slots:
                color, d, alpha, _radii = render_fn(cloud, cameras[frame_ids[slot]], bg)
                keep(slot, color, d, alpha)
                if report and slot + 1 in chunk_ends:
                    on_chunk(slot + 1, rgba[:slot + 1])

    on_gpu = device.type == "cuda"
    # (without a GPU there are no streams to overlap; the pipelined driver still runs there -- same control flow, the
    #  stream hand-overs become no-ops -- which is how the world-size-2 gloo tests exercise it)
    streams = max(1, int(streams)) if (on_gpu or driver == "pipelined") else 1
    if chunk_ends and driver == "threads" and streams > 1 and n >= 2:
        raise ValueError("on_chunk needs the pipelined (or serial) driver")
    if streams == 1 or n < 2:
        render_slots(range(n), report=True)
        for e in late_ends:
            on_chunk(e, rgba[:e])
    elif driver == "pipelined":
        from collections import deque
        import contextlib
        caller = torch.cuda.current_stream(device) if on_gpu else None
        side = side_streams(device, streams) if on_gpu else [None] * streams
        on_stream = (lambda st: torch.cuda.stream(st)) if on_gpu else (lambda st: contextlib.nullcontext())
        for st in side:
            if on_gpu:
                st.wait_stream(caller)             # inputs produced on the caller's stream are visible
        in_flight = deque()

        def finish_oldest():
            slot, st, pending = in_flight.popleft()
            with on_stream(st):
                color, d, alpha, _radii = pending.finish()
                keep(slot, color, d, alpha)
            if slot + 1 in chunk_ends:             # frames finish in order: everything up to `slot` is queued
                for other in side:
                    if on_gpu:
                        caller.wait_stream(other)
                on_chunk(slot + 1, rgba[:slot + 1])

        def oldest_ready():
            r = getattr(in_flight[0][2], "ready", None)
            return r is not None and r()

        with torch.no_grad():
            for slot in range(n):
                # finish what can be finished without waiting (its counters are on the host already) before queueing more
                # first halves; and the oldest frame in any case when every stream holds one (stream slot % streams is its)
                while in_flight and (len(in_flight) == streams or oldest_ready()):
                    finish_oldest()
                st = side[slot % streams]
                with on_stream(st):
                    in_flight.append((slot, st, begin_fn(cloud, cameras[frame_ids[slot]], bg)))
            while in_flight:
                finish_oldest()
        for st in side:
            if on_gpu:
                caller.wait_stream(st)             # results are ordered before later work of the caller
        for e in late_ends:
            on_chunk(e, rgba[:e])
    else:
        import threading
        caller = torch.cuda.current_stream(device)
        side = side_streams(device, streams)

        def worker(t):
            torch.cuda.set_device(device)
            side[t].wait_stream(caller)            # inputs produced on the caller's stream are visible
            with torch.cuda.stream(side[t]):
                render_slots(range(t, n, streams))

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(streams)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for st in side:
            caller.wait_stream(st)                 # results are ordered before later work of the caller
    out = {"rgba8": rgba}
    if keep_depth:
        out["depth"] = depth
    return out


def gather_frames(local: torch.Tensor, num_frames: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """Final gather of a round-robin-sharded stack ``[n_local, ...]`` to ``dst`` in frame order.

    Shards differ in length by at most one frame; every rank pads to the common maximum so a single
    ``gather`` (RCCL: all seven xGMI links into the root active at once) moves everything.  Returns
    ``[num_frames, ...]`` on ``dst`` and ``None`` elsewhere.  With no process group it is the identity.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return local[:num_frames]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (num_frames + world - 1) // world
    padded = local
    if local.shape[0] < per:
        pad = torch.zeros((per - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat((local, pad), dim=0)
    padded = padded.contiguous()
    bufs = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty((num_frames,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        ids = shard_frames(num_frames, r, world)
        if ids:
            out[ids] = bufs[r][:len(ids)]
    return out


def broadcast_cloud(cloud: Optional[GaussianCloud], src: int = 0, device=None, group=None) -> GaussianCloud:
    """Give every rank the Gaussians that only rank ``src`` has loaded: the shapes travel as one
    small object broadcast, the parameters as ONE broadcast of a flat fp32 buffer (C3: 0.71 GB, once per job).
    Without a process group it returns ``cloud`` unchanged."""
    if not (dist.is_available() and dist.is_initialized()):
        return cloud
    rank = dist.get_rank(group)
    names = ("means3D", "opacities", "scales", "rotations", "shs", "colors_precomp")
    meta = [None]
    if rank == src:
        meta = [{"shapes": {n: (None if getattr(cloud, n) is None else tuple(getattr(cloud, n).shape)) for n in names},
                 "sh_degree": int(cloud.sh_degree)}]
    dist.broadcast_object_list(meta, src=src, group=group)
    shapes = meta[0]["shapes"]
    if device is None:
        device = cloud.means3D.device if rank == src else torch.device("cpu")
    sizes = {n: int(torch.Size(sh).numel()) for n, sh in shapes.items() if sh is not None}
    flat = torch.empty(sum(sizes.values()), dtype=torch.float32, device=device)
    if rank == src:
        torch.cat([getattr(cloud, n).detach().to(device=device, dtype=torch.float32).reshape(-1) for n in sizes], out=flat)
    dist.broadcast(flat, src=src, group=group)
    parts, at = {}, 0
    for n in names:
        if shapes[n] is None:
            parts[n] = None
        else:
            parts[n] = flat[at:at + sizes[n]].view(shapes[n])
            at += sizes[n]
    return GaussianCloud(parts["means3D"], parts["opacities"], parts["scales"], parts["rotations"], parts["shs"],
                         parts["colors_precomp"], meta[0]["sh_degree"])


def render_and_gather(cloud: GaussianCloud, cameras: Sequence[Camera], frame_ids: Sequence[int], bg: torch.Tensor,
                      dst: int = 0, render_fn: RenderFn = rasterize, streams: int = DEFAULT_STREAMS, chunks: int = 4,
                      group=None, driver: str = "auto", begin_fn: Optional[BeginFn] = None,
                      rows: Optional[int] = None, stats: Optional[dict] = None) -> Optional[torch.Tensor]:
    """Render ``frame_ids`` (this rank's frames) and gather the RGBA8 frames to ``dst`` while rendering continues: the
    shard is cut into ``chunks`` pieces, and as soon as a piece is rendered its ``gather`` is launched asynchronously
    (RCCL runs it on its own stream over xGMI) behind the next piece's rendering, so only the last piece's transfer is
    left as a tail.  With the pipelined (or serial) driver the frame pipeline is not drained at the piece boundaries;
    with ``driver="threads"`` each piece is its own ``render_shard`` call.

    Shards may differ in length (round-robin over F frames not divisible by the world size): every rank's stack is
    padded with zero frames to ``rows`` = the longest shard (given, or agreed on with one small all-reduce).  Returns
    ``[world, rows, 4, H, W]`` on ``dst`` (rank-major; ``frames_in_order`` puts it into frame order), ``None``
    elsewhere; without a process group ``[1, n, 4, H, W]``.  ``stats`` (optional dict) receives ``render_s`` (host
    time until the last frame and the last piece's gather are queued) and ``gather_tail_s`` (the wait after that)."""
    import time
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    n_own = len(frame_ids)
    if rows is None:
        rows = n_own
        if distributed and world > 1:
            t = torch.tensor([n_own], dtype=torch.int64, device=cloud.means3D.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            rows = int(t.item())
    n = max(int(rows), n_own)
    chunks = max(1, min(int(chunks), n))
    t_start = time.perf_counter()
    bounds = [n * k // chunks for k in range(chunks + 1)]
    parts, works, out = [], [], [None]
    starts = {b: a for a, b in zip(bounds[:-1], bounds[1:])}

    def transfer(part, end):
        parts.append(part)                       # kept alive until the transfers have been waited for
        if not distributed:
            return
        bufs = None
        if rank == dst:                          # pieces land directly in their place of the result: no re-assembly
            if out[0] is None:
                out[0] = torch.empty((world, n) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
            bufs = [out[0][r, starts[end]:end] for r in range(world)]
        works.append(dist.gather(part, bufs, dst=dst, group=group, async_op=True))

    split_available = begin_fn is not None or render_fn is rasterize
    on_gpu = cloud.means3D.device.type == "cuda"
    if driver == "auto" and begin_fn is not None:
        driver = "pipelined"
    threaded = on_gpu and int(streams) > 1 and (driver == "threads" or (driver == "auto" and not split_available))
    if threaded:
        for a, b in zip(bounds[:-1], bounds[1:]):
            transfer(render_shard(cloud, cameras, list(frame_ids[a:b]), bg, False, render_fn, streams, "threads",
                                  pad_to=b - a)["rgba8"], b)
    else:
        render_shard(cloud, cameras, list(frame_ids), bg, False, render_fn, streams, driver, begin_fn,
                     chunk_ends=bounds[1:], on_chunk=lambda end, done: transfer(done[starts[end]:end], end), pad_to=n)
    t_queued = time.perf_counter()
    for w in works:
        w.wait()
    if stats is not None:
        if cloud.means3D.device.type == "cuda":
            torch.cuda.synchronize(cloud.means3D.device)
        stats["render_s"] = t_queued - t_start
        stats["gather_tail_s"] = time.perf_counter() - t_queued
    if not distributed:
        return torch.cat(parts, dim=0)[None]
    return out[0] if rank == dst else None


def frames_in_order(gathered: torch.Tensor, num_frames: int) -> torch.Tensor:
    """``render_and_gather``'s rank-major ``[world, rows, ...]`` result of a round-robin job as ``[num_frames, ...]`` in
    frame order (frame ``f`` is row ``f // world`` of rank ``f % world``); the padding rows are dropped."""
    world, rows = int(gathered.shape[0]), int(gathered.shape[1])
    if num_frames > world * rows:
        raise ValueError(f"{num_frames} frames do not fit {world} x {rows} rows")
    return gathered.transpose(0, 1).reshape((world * rows,) + tuple(gathered.shape[2:]))[:num_frames]


def render_trajectory(cloud: GaussianCloud, cameras: Sequence[Camera], bg: torch.Tensor, keep_depth: bool = False,
                      dst: int = 0, render_fn: RenderFn = rasterize, streams: int = DEFAULT_STREAMS,
                      driver: str = "auto", begin_fn: Optional[BeginFn] = None) -> Optional[Dict[str, torch.Tensor]]:
    """Shard -> render -> gather.  Works with or without an initialised process group."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    ids = shard_frames(len(cameras), rank, world)
    local = render_shard(cloud, cameras, ids, bg, keep_depth, render_fn, streams, driver, begin_fn)
    gathered = {k: gather_frames(v, len(cameras), dst) for k, v in local.items()}
    return gathered if rank == dst else None
