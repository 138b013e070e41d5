#!/usr/bin/env python
"""bench.py -- frames/s of the 3DGS forward render path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one ``GaussianRasterizer.forward`` call (SH colour path) on one frame of the workload's
orbit trajectory, plus the RGBA8 pack of that frame; with N > 1 ranks the frames are dealt
round-robin (frame-parallel, weak scaling: every rank renders K frames) and the packed frames are
gathered to rank 0 with one RCCL gather inside the timed region.  Within a GPU the K frames are
issued on ``--streams`` HIP streams (default 2, one host thread each; 1 = strictly serial calls) so
that one frame's VALU-bound blend overlaps the next frame's HBM-bound projection and sorts; every
frame is still one complete forward call and all K are finished inside the timed region.  Inputs (Gaussians, camera
matrices) are resident in HBM before the clock starts.  Rank 0 prints ONE JSON line.

Workload (default ``c3``) = BASELINE.json configs[2] shape on one GPU: 3 M synthetic Gaussians,
1920x1080, 800-frame orbit, SH degree 3 (M = 16), bg = 0.  ``--workload c2`` is the 1 M / 960x540
stand-in.  Weights are random by construction (there are no checkpoints offline): data = synthetic.

Extra objects on the JSON line:
  roofline      dominant kernel of the frame, measured live with HIP events recorded by the library
                on the launch stream during the timed region; ``frame`` holds the whole-frame figure
                against SURVEY.md section 8d's B_alg.
  cpu_baseline  the CPU oracle (C + OpenMP restatement of the reference kernels, oracle/) timed on
                this box's host cores on a bounded sample (rank 0, N = 1 only), with the parity of
                the GPU frame against it.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

WORKLOADS = {
    "c3": dict(name="C3: 3M synthetic Gaussians, 1920x1080, 800-frame orbit (BASELINE configs[2], per-GPU shard)",
               cfg="config_c3", width=1920, height=1080, frames=800),
    "c2": dict(name="C2 stand-in: 1M synthetic Gaussians, 960x540, 200-frame orbit (BASELINE configs[1])",
               cfg="config_c2", width=960, height=540, frames=200),
}


def algorithmic_bytes(P, V, D, T, W, H, M=16):
    """SURVEY.md section 8d, per stage (bytes per frame)."""
    bit = max(1, int(T).bit_length())
    n_pass = -(-(32 + bit) // 8)
    st = {
        "preprocess": 20 * P + (32 + 12 * M + 40) * V,
        "scan": 8 * P,
        "duplicate": 20 * V + 12 * D,
        "sort": 24 * D * n_pass,
        "ranges": 8 * D + 8 * T,
        "blend": 44 * D + 24 * W * H,
    }
    st["frame"] = sum(st.values())
    st["n_pass"] = n_pass
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--gaussians", type=int, default=0, help="override P (parity/debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-hip", action="store_true",
                    help="skip timing the reference's own kernels compiled for gfx950 (oracle/_ref/libgsr_ref_hip.so)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of the frames to rank 0 (N > 1)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="debug: take the N > 1 code path (process group, barriers, pipelined RCCL gather) even with "
                         "one rank, so that path can be exercised on a one-GPU box")
    ap.add_argument("--gather-chunks", type=int, default=16,
                    help="N > 1: pieces the per-rank frame stack is gathered in, each overlapped with the next piece's rendering")
    ap.add_argument("--streams", type=int, default=3,
                    help="render frames on this many HIP streams so independent frames overlap")
    ap.add_argument("--driver", choices=["auto", "pipelined", "threads"], default="auto",
                    help="how the streams are fed: 'pipelined' = one host thread, each call split where the host needs "
                         "the pair count; 'threads' = one blocking host thread per stream; auto = pipelined")
    ap.add_argument("--boundary", choices=["op", "render"], default="op",
                    help="op: one GaussianRasterizer.forward per frame (the headline); render: the reference's "
                         "whole per-frame render() = activations + SH pass + normal pass + normal post-processing")
    ap.add_argument("--no-geometry-cache", action="store_true", help="A/B knob: recompute geometry for the 2nd pass")
    ap.add_argument("--no-cull", action="store_true", help="A/B knob: GSR_OPT_TILE_CULL = 0")
    ap.add_argument("--slabs", type=int, default=None, help="A/B knob: GSR_OPT_SLABS (1 = no depth slabs; default: as the scene calls for)")
    ap.add_argument("--slab-first", type=int, default=None, help="A/B knob: GSR_OPT_SLAB_FIRST (pairs per tile in the first slab)")
    ap.add_argument("--no-defer-colour", action="store_true", help="A/B knob: GSR_OPT_DEFER_COLOUR = 0")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout when a process
    # group is created (seen with RCCL 2.26), so the real stdout is set aside for that line and everything else
    # that writes to file descriptor 1 lands on stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    import torch.distributed as dist
    distributed = world > 1 or args.force_distributed
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.frame_parallel import pack_rgba8, rasterize, rasterize_begin, render_and_gather, side_streams
    from diff_gaussian_rasterization import _C

    if args.no_cull:
        _lib.set_option(_lib.OPT_TILE_CULL, 0)
    if args.slabs is not None:
        _lib.set_option(_lib.OPT_SLABS, args.slabs)
    if args.slab_first is not None:
        _lib.set_option(_lib.OPT_SLAB_FIRST, args.slab_first)
    if args.no_defer_colour:
        _lib.set_option(_lib.OPT_DEFER_COLOUR, 0)
    wl = WORKLOADS[args.workload]
    W, H, F = wl["width"], wl["height"], wl["frames"]
    cfg = getattr(scenes, wl["cfg"])
    cloud_cpu = cfg(P=args.gaussians) if args.gaussians else cfg()
    cloud = cloud_cpu.to(device)
    cams_cpu = orbit_cameras(F, W, H)
    K, Wm = args.steps, args.warmup
    # rank r renders frames r, r+N, ... ; warm-up frames precede the timed ones on the same orbit
    frame_of = lambda i: (i * world + rank) % F
    need = sorted({frame_of(i) for i in range(Wm + K)})
    cams = {f: cams_cpu[f].to(device) for f in need}
    bg = torch.zeros(3, dtype=torch.float32, device=device)
    rgba = torch.empty((K, 4, H, W), dtype=torch.uint8, device=device)
    P, M = cloud.P, int(cloud.shs.shape[1])
    T = ((W + 15) // 16) * ((H + 15) // 16)

    if args.no_geometry_cache:
        _C.set_geometry_cache(False)
    if args.boundary == "render":
        from autovfx_amd import renderer
        from autovfx_amd.gaussian_model import GaussianModel
        model = GaussianModel.from_activated(cloud.means3D, cloud.opacities, cloud.scales, cloud.rotations, cloud.shs,
                                             cloud.sh_degree)

        def render_fn_boundary(_cloud, cam, bg_):
            out = renderer.render(cam, model, renderer.PipelineParams, bg_)
            return out["render"][:3], out["depth"][None], out["render"][3:4], out["radii"]

        class _PendingBoundary:
            def __init__(self, pending):
                self.pending = pending

            def finish(self):
                out = self.pending.finish()
                return out["render"][:3], out["depth"][None], out["render"][3:4], out["radii"]

        def begin_fn(_cloud, cam, bg_):
            return _PendingBoundary(renderer.render_begin(cam, model, renderer.PipelineParams, bg_))

        def step(i, slot):
            out = renderer.render(cams[frame_of(i)], model, renderer.PipelineParams, bg)
            pack_rgba8(out["render"][:3], out["render"][3:4], out=rgba[slot % K])
    else:
        begin_fn = rasterize_begin

        def step(i, slot):
            color, _depth, alpha, _radii = rasterize(cloud, cams[frame_of(i)], bg)
            pack_rgba8(color, alpha, out=rgba[slot % K])

    S = max(1, args.streams)
    streams = side_streams(device, S) if S > 1 else []   # the driver's own: allocator pools warmed here stay warm there
    driver = args.driver
    if driver == "auto":
        driver = "pipelined"

    def run_steps(first, count):
        """Steps first .. first+count-1; with S > 1 streams, step j goes to host thread / stream j % S."""
        if S == 1:
            for j in range(count):
                step(first + j, j)
            return
        if driver == "pipelined":   # one host thread; frame j's second half is queued after frame j+S-1's first half
            from collections import deque
            in_flight = deque()

            def finish_oldest():
                j, st, pending = in_flight.popleft()
                with torch.cuda.stream(st):
                    color, _depth, alpha, _radii = pending.finish()
                    pack_rgba8(color, alpha, out=rgba[j % K])

            with torch.no_grad():
                for j in range(count):
                    if len(in_flight) == S:
                        finish_oldest()
                    st = streams[j % S]
                    with torch.cuda.stream(st):
                        in_flight.append((j, st, begin_fn(cloud, cams[frame_of(first + j)], bg)))
                while in_flight:
                    finish_oldest()
            for st in streams:
                torch.cuda.current_stream(device).wait_stream(st)
            return
        import threading

        def worker(t):
            torch.cuda.set_device(device)
            with torch.no_grad(), torch.cuda.stream(streams[t]):
                for j in range(t, count, S):
                    step(first + j, j)

        threads = [threading.Thread(target=worker, args=(t,)) for t in range(S)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for st in streams:   # later work on the caller's stream sees every frame
            torch.cuda.current_stream(device).wait_stream(st)

    with torch.no_grad():
        run_steps(0, Wm)
        if S > 1 and Wm < 2 * S + 2:   # every stream's allocator pool and scratch sizes settle before the clock starts
            run_steps(Wm, min(K, 2 * S + 2 - Wm))   # untimed; these frames are rendered again inside the timed region
        if distributed and not args.no_gather:
            # the gather path's own warm-up: the frame stack, the receive buffers and RCCL's first gather are
            # allocated / initialised by an untimed rehearsal of the same call (its frames are rendered again below)
            cam_list = [cams[frame_of(Wm + j)] for j in range(K)]
            render_and_gather(cloud, cam_list, list(range(K)), bg, dst=0, streams=S, chunks=args.gather_chunks,
                              driver=driver, begin_fn=begin_fn,
                              render_fn=(render_fn_boundary if args.boundary == "render" else rasterize))
        _lib.set_stage_timing(S == 1)   # per-stage events are per host thread; only read them single-stream
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        t0 = time.perf_counter()
        gathered = None
        if distributed and not args.no_gather:
            # N > 1: the same per-frame work through the frame-parallel driver, whose gather to rank 0 is cut into
            # pieces that travel over xGMI behind the rendering of the next piece (only the last one is a tail)
            gathered = render_and_gather(cloud, cam_list, list(range(K)), bg, dst=0, streams=S, chunks=args.gather_chunks,
                                         driver=driver, begin_fn=begin_fn,
                                         render_fn=(render_fn_boundary if args.boundary == "render" else rasterize))
        else:
            run_steps(Wm, K)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if S > 1:   # stage breakdown from an untimed single-stream replay of the first frames
            _lib.set_stage_timing(True)
            for j in range(min(K, 16)):
                step(Wm + j, j)
            torch.cuda.synchronize()
        stage_ms = _lib.stage_times_ms()
        call_ms = sorted(_lib.call_times_ms())
        _lib.set_stage_timing(False)

    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- per-frame workload statistics (untimed): V and D of this rank's timed frames ----
    Vs, Ds = [], []
    sample = list(range(0, K, max(1, K // 8)))
    with torch.no_grad():
        e = torch.Tensor([])
        for i in sample:
            cam = cams[frame_of(Wm + i)]
            n, _c, _d, _a, radii, *_ = _C.rasterize_gaussians(
                bg, cloud.means3D, e, cloud.opacities, cloud.scales, cloud.rotations, 1.0, e,
                cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, cloud.shs,
                cloud.sh_degree, cam.camera_center, False, False)
            Ds.append(int(n))
            Vs.append(int((radii > 0).sum().item()))
    V, D = float(np.mean(Vs)), float(np.mean(Ds))
    alg = algorithmic_bytes(P, V, D, T, W, H, M)

    fps = K * world / elapsed
    ms_per_step = elapsed / K * 1e3

    calls = stage_ms.pop("calls")
    stage_alg = {"preprocess": alg["preprocess"], "depth_sort": 0, "scan": alg["scan"], "duplicate": alg["duplicate"],
                 "tile_sort": alg["sort"], "ranges": alg["ranges"], "blend": alg["blend"], "colour": 0}
    stages = {k: {"ms": round(v, 4), "alg_bytes": int(stage_alg[k]),
                  "alg_GBps": round(stage_alg[k] / (v * 1e-3) / 1e9, 1) if v > 0 else None}
              for k, v in stage_ms.items()}
    dom = max(stage_ms, key=stage_ms.get)
    kernel_of = {"blend": "blend_quadrant_kernel", "preprocess": "preprocess_kernel", "duplicate": "expand_kernel",
                 "ranges": "tile_ranges_kernel", "colour": "sh_colour_kernel"}
    traffic, traffic_src = pmc_traffic(kernel_of.get(dom, dom)) if (args.workload == "c3" and not args.gaussians) else (None, None)
    dom_gbps = stage_alg[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    frame_gbps = alg["frame"] / (ms_per_step * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "kernel": kernel_of.get(dom, dom),
        "achieved": round(dom_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(dom_gbps / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
        "avg_launch_ms": round(stage_ms[dom], 4), "alg_bytes_per_launch": int(stage_alg[dom]),
        "timed_calls": calls,
        "frame": {"alg_bytes": int(alg["frame"]), "achieved": round(frame_gbps, 1),
                  "frac": round(frame_gbps / HBM_PEAK_GBPS, 4), "n_pass_ref_sort": alg["n_pass"],
                  # one frame alone on the GPU, first kernel to last (HIP events), over the per-stage replay
                  "single_stream_ms_p50": round(call_ms[len(call_ms) // 2], 4) if call_ms else None,
                  "single_stream_ms_p95": round(call_ms[min(len(call_ms) - 1, int(0.95 * len(call_ms)))], 4) if call_ms else None},
        "stages": stages,
    }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(cloud_cpu, cams_cpu, frame_of(Wm), cloud, cams, bg, W, H)
    reference_on_gpu = None
    if rank == 0 and world == 1 and args.boundary == "op" and not args.no_reference_hip:
        reference_on_gpu = run_reference_on_gpu(cloud, [cams[frame_of(Wm + j)] for j in range(min(K, 24))], bg)

    if rank == 0:
        line = {
            "metric": ("rendered frames/sec at 1920x1080, 3M Gaussians" if args.workload == "c3" and not args.gaussians
                       else f"rendered frames/sec at {W}x{H}, {P} Gaussians")
                      + (" [render() boundary]" if args.boundary == "render" else ""),
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["name"], "P": P, "sh_coeffs": M, "width": W, "height": H, "tiles": T,
                       "visible_mean": round(V, 1), "num_rendered_mean": round(D, 1),
                       "boundary": ("render(): activations, SH pass, normal pass (geometry reused), normal/pseudo-normal "
                                    "post-processing, RGBA8 pack per frame" if args.boundary == "render" else
                                    "GaussianRasterizer.forward (SH) + RGBA8 pack per frame")
                                   + ("; RCCL gather of the RGBA8 frames to rank 0, pipelined behind the rendering" if distributed and not args.no_gather else ""),
                       "parallelism": f"frame-parallel x{world}", "streams_per_gpu": S, "stream_driver": driver if S > 1 else "serial",
                       "options": {"tile_cull": _lib.get_option(_lib.OPT_TILE_CULL),
                                   "slabs": _lib.get_option(_lib.OPT_SLABS), "slab_first": _lib.get_option(_lib.OPT_SLAB_FIRST),
                                   "defer_colour": _lib.get_option(_lib.OPT_DEFER_COLOUR)}},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if reference_on_gpu is not None:
            line["reference_on_gpu"] = reference_on_gpu
        if gathered is not None:
            line["config"]["gathered_frames"] = int(gathered.shape[0] * gathered.shape[1])
            line["config"]["gather_chunks"] = args.gather_chunks
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


def pmc_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/): FETCH_SIZE and
    WRITE_SIZE are KiB counted at the L2's memory side; on gfx950 FETCH_SIZE reports half of a wide coalesced
    read stream, hence the factor 2 (/opt/skills/guides/MI355X_MICROARCH.md, HBM).  None if no profile exists."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_per_kernel_mean.csv")
    try:
        vals = {}
        with open(path) as f:
            for line in f:
                parts = line.strip().rsplit(",", 3)   # template arguments in the kernel name may hold commas
                if len(parts) != 4 or parts[1] not in ("FETCH_SIZE", "WRITE_SIZE"):
                    continue
                name = parts[0]
                if name == kernel or (name.split("<")[0] == kernel and name.endswith("<false>")):   # the plain variant
                    vals[parts[1]] = float(parts[2])
        if len(vals) == 2:
            return int((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024), "profiles/r01_pmc_per_kernel_mean.csv (2*FETCH_SIZE + WRITE_SIZE, KiB)"
    except OSError:
        pass
    return None, None


def run_reference_on_gpu(cloud, cam_list, bg):
    """The REFERENCE's own rasterizer (its CUDA sources compiled for gfx950 by oracle/build_ref_hip.py, a measuring
    stick) on the same frames of the same workload on this GPU: what "matching the reference" means here.  One stream,
    one frame at a time, as the reference runs; its scratch arenas are kept between frames."""
    try:
        from oracle import ref_hip
        if not ref_hip.available():
            return None
        dev = cloud.means3D.device
        H, W = int(cam_list[0].image_height), int(cam_list[0].image_width)
        outs = (torch.zeros((3, H, W), device=dev), torch.zeros((1, H, W), device=dev), torch.zeros((1, H, W), device=dev),
                torch.zeros((cloud.P,), dtype=torch.int32, device=dev))
        for cam in cam_list[:3]:
            ref_hip.forward(cloud, cam, bg, outs)
        t0 = time.perf_counter()
        for cam in cam_list:
            ref_hip.forward(cloud, cam, bg, outs)
        dt = time.perf_counter() - t0
        return {"value": round(len(cam_list) / dt, 2), "unit": "frames/s", "ms_per_frame": round(dt / len(cam_list) * 1e3, 3),
                "kind": "reference sources (forward.cu, rasterizer_impl.cu; hipCUB for its two CUB calls) compiled for gfx950, "
                        "-ffp-contract=off", "sample": f"{len(cam_list)} frames of the same workload, one stream"}
    except Exception as e:  # a measuring stick must not break the headline line
        return {"error": repr(e)[:200]}


def torch_cpu_c1(threads: int = 16, timeout_s: float = 30.0):
    """The PyTorch-CPU restatement of the splat (oracle/torch_splat.py, BASELINE configs[0]: 10 k Gaussians, one
    256x256 camera) timed on the same host: the "CPU-only PyTorch splat path" figure, at the one size it finishes in
    well under a second.  Runs in a subprocess with a hard time limit and a bounded thread count (with one thread per
    core of a 256-thread host its many small tensor ops take minutes, not seconds)."""
    import subprocess
    code = (
        "import sys, time, json, torch; sys.path.insert(0, %r); torch.set_num_threads(%d)\n"
        "from oracle import torch_splat; from autovfx_amd import scenes\n"
        "cloud, cam = scenes.config_c1(), scenes.c1_camera()\n"
        "kw = dict(means3D=cloud.means3D, opacities=cloud.opacities, width=cam.image_width, height=cam.image_height,"
        " viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,"
        " tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, sh_degree=cloud.sh_degree, scale_modifier=1.0, shs=cloud.shs,"
        " scales=cloud.scales, rotations=cloud.rotations)\n"
        "best = 1e9\n"
        "for _ in range(2):\n"
        "    t0 = time.perf_counter(); torch_splat.forward(bg=torch.zeros(3), **kw); best = min(best, time.perf_counter() - t0)\n"
        "print(json.dumps({'s': best}))\n") % (ROOT, threads)
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s)
        best = json.loads(r.stdout.strip().splitlines()[-1])["s"]
        return {"value": round(1.0 / best, 3), "unit": "frames/s", "cores": threads,
                "sample": "C1 (10k Gaussians, 256x256), oracle/torch_splat.py, best of 2"}
    except Exception as e:  # the headline line must not depend on this extra
        return {"error": repr(e)[:200]}


def host_cpu_budget():
    """(logical CPUs, CPUs this container may actually use): a cgroup quota caps the second (cpu.max "quota period")."""
    n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return n, min(float(n), float(quota) / float(period))
    except (OSError, ValueError):
        pass
    return n, float(n)


def run_cpu_baseline(cloud_cpu, cams_cpu, frame, cloud, cams, bg, W, H, budget_s=12.0, max_frames=12):
    """Time the CPU oracle (OpenMP over the CPUs this container may use) on a bounded sample of the same workload --
    frames of the same orbit until ~budget_s of wall time -- and report the GPU's parity on the first of them."""
    from oracle import cpu_oracle
    from autovfx_amd.frame_parallel import rasterize
    logical, usable = host_cpu_budget()
    threads = max(1, int(round(usable)))
    try:   # one OpenMP thread per usable CPU: 256 threads inside a 16-CPU quota only fight each other
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        threads = logical

    def kw(cam):
        return dict(means3D=cloud_cpu.means3D, opacities=cloud_cpu.opacities, bg=np.zeros(3, np.float32), width=W,
                    height=H, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                    campos=cam.camera_center, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                    sh_degree=cloud_cpu.sh_degree, shs=cloud_cpu.shs, scales=cloud_cpu.scales,
                    rotations=cloud_cpu.rotations)

    cpu_oracle.lib()
    ref, total, n = None, 0.0, 0
    F = len(cams_cpu)
    while n < max_frames and total < budget_s:
        t0 = time.perf_counter()
        out = cpu_oracle.forward(**kw(cams_cpu[(frame + 7 * n) % F]))
        total += time.perf_counter() - t0
        ref = out if ref is None else ref
        n += 1
    with torch.no_grad():
        color, depth, alpha, radii = rasterize(cloud, cams[frame], bg)
    torch.cuda.synchronize()
    err = np.abs(color.cpu().numpy() - ref["color"]).max(axis=0)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": round(n / total, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} frames of the same workload at full size (orbit indices {frame}+7k), C+OpenMP oracle, "
                      f"{total:.2f} s of wall time on {threads} OpenMP threads ({logical} logical CPUs, "
                      f"cgroup quota {usable:g} CPUs)",
            "cpu_model": model, "torch_cpu_c1": torch_cpu_c1(),
            "parity": {"frame": frame, "rgb_maxabs": float(err.max()), "rgb_px_over_1e-4": int((err > 1e-4).sum()),
                       "alpha_maxabs": float(np.abs(alpha.cpu().numpy() - ref["alpha"]).max()),
                       "depth_maxabs": float(np.abs(depth.cpu().numpy() - ref["depth"]).max()),
                       "radii_equal": bool((radii.cpu().numpy() == ref["radii"]).all()),
                       "pixels": int(W * H)}}


if __name__ == "__main__":
    main()
