#!/usr/bin/env python
"""bench.py -- frames/s of the 3DGS forward render path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A *step* is one ``GaussianRasterizer.forward`` call (SH colour path, an inference call: nothing requires a gradient)
on one frame of the workload's orbit trajectory, plus the RGBA8 pack of that frame.  Inputs (Gaussians, camera
matrices) are resident in HBM before the clock starts.  Within a GPU the K frames are issued on ``--streams`` HIP
streams (default 11, on the runtime's 4 hardware queues) by ONE host thread that splits every call where the host needs the pair count (``--driver
pipelined``), so that one frame's VALU-bound blend overlaps other frames' HBM- and latency-bound stages; every frame is
still one complete forward call and all K are finished inside the timed region.

The timed region -- W untimed warm-up steps once, then EXACTLY K steps between ``barrier + synchronize`` on both
sides, max over ranks -- is repeated ``--regions`` times (default 5): ``value`` / ``ms_per_step`` are the MEDIAN
region, ``regions`` lists them all.

With N > 1 ranks (weak scaling, the default): every rank renders K frames dealt round-robin and the packed frames are
gathered to rank 0 with RCCL inside the timed region, in pieces that travel behind the rendering of the next piece.
``--job-frames F`` instead times ONE fixed job (strong scaling; BASELINE configs[2] is 800 frames on 8 GPUs): rank 0
owns the cloud, one broadcast, F frames dealt round-robin (shards may differ by one frame), pipelined gather.

Workload (default ``c3``) = BASELINE.json configs[2] shape on one GPU: 3 M synthetic Gaussians, 1920x1080, 800-frame
orbit, SH degree 3 (M = 16), bg = 0.  Other workloads (``--workload``; short runs of them also ride on the default
line under ``also``): ``c2`` 1 M / 960x540; ``heavy`` a trained-scene-like stress cloud (heavy-tailed sizes, 10:1
anisotropy, bimodal opacity); ``c4`` 200 k flat SuGaR-style Gaussians, RGB + normal + depth in one fused call;
``c5`` C2 frames -> RGBA8 -> composited with synthetic Blender layers.  Data is synthetic by construction.

Extra objects on the JSON line:
  roofline      dominant kernel of the frame (the blend): algorithmic bytes per launch / average launch time, measured
                with HIP events recorded by the library on the launch stream in an untimed single-stream replay right
                after the timed regions (``stage_replay_calls`` frames; rocprofv3 of ``--streams 1`` agrees);
                ``traffic`` = HBM bytes per launch from the committed PMC passes (profiles/, stamped with the commit they
                were taken at); ``frame`` = the whole frame against SURVEY.md 8d's B_alg and against counter traffic.
  cpu_baseline  the CPU oracle (C + OpenMP restatement of the reference kernels, oracle/) and the PyTorch-CPU splat
                (oracle/torch_splat.py) timed on this box's host cores on a bounded sample (rank 0, N = 1 only), with
                the parity of the GPU frames against the oracle (first / middle / last frame of the orbit).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
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
    "heavy": dict(name="heavy: 1M trained-scene-like Gaussians (log-normal sizes sigma 1.2, 10:1 anisotropy, bimodal "
                       "opacity), 960x540, 200-frame orbit (stress workload, not a BASELINE config)",
                  cfg="config_heavy", width=960, height=540, frames=200),
    "heavy1080": dict(name="heavy at 1920x1080 (stress workload, not a BASELINE config)",
                      cfg="config_heavy", width=1920, height=1080, frames=200),
    "c4": dict(name="C4: 200k flat SuGaR-style Gaussians, colors_precomp, SuGaR's off-centre principal-point projection, "
                    "RGB + normal + depth in one fused call, 960x540, 50-frame orbit (BASELINE configs[3])",
               cfg="config_c4", width=960, height=540, frames=50),
    "c5": dict(name="C5: C2 frames -> RGBA8 -> composite with synthetic Blender layers (object, shadow catcher, 3DGS "
                    "object, smoke + fire), 960x540, 400 frames (BASELINE configs[4])",
               cfg="config_c2", width=960, height=540, frames=400),
}


def algorithmic_bytes(P, V, D, T, W, H, M=16):
    """SURVEY.md section 8d, per stage (bytes per frame)."""
    bit = max(1, int(T).bit_length())
    n_pass = -(-(32 + bit) // 8)
    vis = (32 + 12 * M + 40) if M else (32 + 12 + 40)   # colors_precomp: 12 B read instead of the SH record
    st = {
        "preprocess": 20 * P + vis * V,
        "scan": 8 * P,
        "duplicate": 20 * V + 12 * D,
        "sort": 24 * D * n_pass,
        "ranges": 8 * D + 8 * T,
        "blend": 44 * D + 24 * W * H,
    }
    st["frame"] = sum(st.values())
    st["n_pass"] = n_pass
    return st


def design_bytes(u, inference=True, deferred=True):
    """What THIS library's stages have to move per frame at the least ("design bytes"): per-unit
    figures of its own data layout x the measured units of the frame.  u: P Gaussians, V visible (radii > 0), E emitting
    (a non-empty tight rectangle), pairs = expanded pairs per depth slab, L = Gaussians whose colour is evaluated, T tiles,
    W x H pixels, M SH coefficients.  Lower bounds: the re-scan of a later slab and the parking of unfinished pixels between
    two blend launches are left out."""
    P, V, E, L, T, W, H, M = u["P"], u["V"], u["E"], u["L"], u["T"], u["W"], u["H"], u["M"]
    pairs, S = float(sum(u["pairs"])), max(1, len(u["pairs"]))
    sh = (12 * M + 12) if M else 12                        # SH record (or precomputed colour) read + xyz / nothing
    st = {
        # xyz 12 r; radii 4, splat record 16, depth key 4, listed byte 1 w | scale 12 + quat 16 + opacity 4 r, raster record 32 w
        "preprocess": 37 * P + 64 * V + (0 if (inference and deferred and M) else (sh + 12) * V),
        # 4 passes: histogram reads 4; scatter reads key (+ payload past the first pass), writes payload (+ key before the last)
        "depth_sort": 72 * P,
        "scan": 0,
        # gather: order 4 + record 16 r, record 16 + offset 4 w per emitter; offsets r/w 8 per Gaussian; expansion: offset 4 +
        # record 16 + id 4 r per emitter, key 4 + id 4 w per pair
        "duplicate": 64 * E + 8 * P + 8 * pairs,
        "tile_sort": 40 * pairs,                           # 2 passes x (4 histogram + 8 r + 8 w)
        "ranges": 8 * T * S,
        # SH 12 M + xyz 12 r, colour 12 w per evaluated Gaussian; the listed bytes once per launch
        "colour": ((12 * M + 24) * L + P * S) if (inference and deferred and M) else 0,
        # list entry 4 r per pair; raster 32 + colour 12 r once per listed Gaussian; colour 12 + depth 4 + alpha 4 + n_contrib 4 w
        "blend": 4 * pairs + 44 * L + 24 * W * H,
    }
    st["frame"] = sum(st.values())
    return st


# kernels of each stage (names as rocprofv3 prints them, namespaces stripped) for the counter traffic per stage
STAGE_KERNELS = {
    "preprocess": ("preprocess_kernel", "counter_tally_kernel"),
    "depth_sort": (),   # split by call index below: the first 4 passes of a frame are the depth sort
    "duplicate": ("bin_gather_kernel", "bin_offsets_kernel", "slab_bounds_kernel", "slab_recount_kernel", "slab_compact_kernel", "expand_kernel"),
    "tile_sort": (),
    "ranges": ("tile_ranges_kernel",),
    "colour": ("sh_colour_listed_kernel", "sh_colour_all_kernel"),
    "blend": ("blend_quadrant_kernel",),
}


# ---------------------------------------------------------------------------------------------------------------------
# one workload on this rank's GPU
# ---------------------------------------------------------------------------------------------------------------------
class Bench:
    """A workload resident on the GPU and the ways of pushing frames through it."""

    def __init__(self, key, device, args, boundary="op", gaussians=0):
        from autovfx_amd import scenes
        from autovfx_amd.cameras import orbit_cameras
        self.key, self.device, self.boundary = key, device, boundary
        wl = WORKLOADS[key]
        self.name, self.W, self.H, self.F = wl["name"], wl["width"], wl["height"], wl["frames"]
        cfg = getattr(scenes, wl["cfg"])
        self.cloud_cpu = cfg(P=gaussians) if gaussians else cfg()
        self.cloud = self.cloud_cpu.to(device)
        if key == "c4":   # SuGaR's camera: projection matrix with the principal-point terms (sugar_model.py:2029-2030)
            from autovfx_amd.cameras import sugar_orbit_cameras
            self.cams_cpu = sugar_orbit_cameras(self.F, self.W, self.H)
        else:
            self.cams_cpu = orbit_cameras(self.F, self.W, self.H)
        self._cams = {}
        self.bg = torch.zeros(3, dtype=torch.float32, device=device)
        self.P = self.cloud.P
        self.M = int(self.cloud.shs.shape[1]) if self.cloud.shs is not None else 0
        self.T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        self.extra = None
        self.layers = None
        self.model = None
        if key == "c4":   # the per-Gaussian normals SuGaR renders beside the colours (sugar_model.py:2174-2183)
            n = self.cloud.means3D / self.cloud.means3D.norm(dim=1, keepdim=True)
            self.extra = (n * 0.5 + 0.5).contiguous()
        if key == "c5":
            self.layers = synthetic_blender_layers(self.W, self.H, device)
        if boundary == "render":
            from autovfx_amd.gaussian_model import GaussianModel
            c = self.cloud
            self.model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, c.sh_degree)

    def cam(self, f):
        f %= self.F
        if f not in self._cams:
            self._cams[f] = self.cams_cpu[f].to(self.device)
        return self._cams[f]

    # -- the per-frame call, in split form: begin(frame) -> pending; finish(pending) -> (color, depth, alpha, radii) --
    def begin(self, f):
        cam = self.cam(f)
        if self.boundary == "render":
            from autovfx_amd import renderer
            return renderer.render_begin(cam, self.model, renderer.PipelineParams, self.bg)
        if self.extra is not None:
            from diff_gaussian_rasterization import _C
            from autovfx_amd.frame_parallel import settings_for_camera
            s = settings_for_camera(cam, self.bg, self.cloud.sh_degree)
            e = torch.empty(0, dtype=torch.float32, device=self.device)
            return _C.rasterize_gaussians_begin(
                s.bg, self.cloud.means3D, self.cloud.colors_precomp, self.cloud.opacities, self.cloud.scales,
                self.cloud.rotations, s.scale_modifier, e, s.viewmatrix, s.projmatrix, s.tanfovx, s.tanfovy,
                s.image_height, s.image_width, e, s.sh_degree, s.campos, s.prefiltered, s.debug, self.extra, inference=True)
        from autovfx_amd.frame_parallel import rasterize_begin
        return rasterize_begin(self.cloud, cam, self.bg)

    def finish(self, pending):
        out = pending.finish()
        if self.boundary == "render":
            return out["render"][:3], out["depth"][None], out["render"][3:4], out["radii"]
        if self.extra is not None:
            _n, color, depth, alpha, radii, _g, _b, _i, normal = out
            return color, depth, alpha, radii
        return out

    def consume(self, slot, result, rgba, composed=None):
        from autovfx_amd.frame_parallel import pack_rgba8
        color, _depth, alpha, _radii = result
        pack_rgba8(color, alpha, out=rgba[slot])
        if self.layers is not None:   # C5: the frame is the background layer of the composite (blend_all.py:236-300)
            from autovfx_amd.compositor import composite_frame
            L = self.layers
            bg_c = rgba[slot].permute(1, 2, 0).contiguous()   # planar RGBA8 -> the interleaved layout image files have
            composite_frame(bg_c, L["o_c"], L["o_d"], L["s_c"], L["s_d"], L["o_s_c"], L["o_gs_c"], L["o_gs_d"],
                            L["s_f_c"], L["s_f_d"], L["s_f_c_pre"], out=composed[slot])

    def run(self, frames, rgba, streams, side, composed=None):
        """Render `frames` (orbit indices) into rgba[0..]; S > 1: one host thread keeps S split calls in flight."""
        with torch.no_grad():
            if streams <= 1:
                for j, f in enumerate(frames):
                    self.consume(j % rgba.shape[0], self.finish(self.begin(f)), rgba, composed)
                return
            from collections import deque
            in_flight = deque()

            def finish_oldest():
                j, st, pending = in_flight.popleft()
                with torch.cuda.stream(st):
                    self.consume(j % rgba.shape[0], self.finish(pending), rgba, composed)

            def oldest_ready():
                r = getattr(in_flight[0][2], "ready", None)
                return r is not None and r()

            for j, f in enumerate(frames):
                while in_flight and (len(in_flight) == streams or oldest_ready()):   # (as frame_parallel.render_shard does)
                    finish_oldest()
                st = side[j % streams]
                with torch.cuda.stream(st):
                    in_flight.append((j, st, self.begin(f)))
            while in_flight:
                finish_oldest()
            for st in side:
                torch.cuda.current_stream(self.device).wait_stream(st)

    def units(self, frames):
        """Mean per-frame units of `frames` (untimed inference calls, scratch decoded): V visible (radii > 0), D the
        reference's num_rendered, E emitting splats, L Gaussians whose colour is evaluated, pairs expanded per depth slab."""
        from diff_gaussian_rasterization import _C
        acc = {"V": [], "D": [], "E": [], "L": []}
        pairs = []
        e = torch.Tensor([])
        c = self.cloud
        _C.set_geometry_cache(False)
        try:
            with torch.no_grad():
                for f in frames:
                    cam = self.cam(f)
                    n, _c, _d, _a, radii, geom, *_ = _C.rasterize_gaussians(
                        self.bg, c.means3D, e if c.colors_precomp is None else c.colors_precomp, c.opacities, c.scales,
                        c.rotations, 1.0, e, cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                        self.H, self.W, e if c.shs is None else c.shs, c.sh_degree, cam.camera_center, False, False,
                        inference=True)
                    lay = _C.last_layout()
                    g = lay["geom"]
                    bins = geom[g["splat_bins"]:g["splat_bins"] + 16 * self.P].view(torch.int32).view(-1, 4)
                    E = int((bins[:, 1] != 0).sum().item())
                    L = E
                    if g.get("listed", 0):
                        L = int((geom[g["listed"]:g["listed"] + self.P] != 0).sum().item())
                    acc["D"].append(int(n)); acc["V"].append(int((radii > 0).sum().item())); acc["E"].append(E); acc["L"].append(L)
                    pairs.append(list(lay["slab_pairs"]))
        finally:
            _C.set_geometry_cache(None)
        S = max(len(p) for p in pairs)
        mean_pairs = [float(np.mean([p[k] if k < len(p) else 0 for p in pairs])) for k in range(S)]
        out = {k: float(np.mean(v)) for k, v in acc.items()}
        out.update(P=self.P, T=self.T, W=self.W, H=self.H, M=self.M, pairs=mean_pairs)
        return out

    def stats(self, frames):
        u = self.units(frames)
        return u["V"], u["D"]

    def run_blocking(self, frames, rgba):
        """The loop an unmodified caller runs (scene_representation.py:355-424): one frame at a time on the current stream,
        each a blocking ``GaussianRasterizer.forward`` (or the whole ``render()``), nothing in flight beside it."""
        from autovfx_amd.frame_parallel import rasterize
        with torch.no_grad():
            for j, f in enumerate(frames):
                cam = self.cam(f)
                if self.boundary == "render":
                    from autovfx_amd import renderer
                    out = renderer.render(cam, self.model, renderer.PipelineParams, self.bg)
                    res = (out["render"][:3], out["depth"][None], out["render"][3:4], out["radii"])
                else:
                    res = rasterize(self.cloud, cam, self.bg)
                self.consume(j % rgba.shape[0], res, rgba)


def synthetic_blender_layers(W, H, device, seed=11):
    """The layers blend_all.py reads from Blender's output folders, synthetic and resident: an object pass, a shadow
    catcher pass and their union, a re-rendered 3DGS object, smoke + premultiplied fire; RGBA8 [H,W,4] + fp32 depth."""
    from autovfx_amd.compositor import smoke_depth_fill
    g = torch.Generator(device="cpu").manual_seed(seed)
    yy, xx = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")

    def blob(cx, cy, r):
        d2 = ((xx - cx * W) / (r * W)) ** 2 + ((yy - cy * H) / (r * W)) ** 2
        return torch.clamp(1.5 - d2, 0.0, 1.0)

    def rgba(alpha, tint):
        rgb = torch.rand(H, W, 3, generator=g) * 0.3 + torch.tensor(tint).view(1, 1, 3) * 0.7
        return torch.cat((rgb * 255.0, alpha[..., None] * 255.0), dim=2).clamp(0, 255).to(torch.uint8)

    a_obj, a_gs, a_smoke = blob(0.45, 0.55, 0.12), blob(0.62, 0.5, 0.08), blob(0.5, 0.35, 0.2) * 0.6
    far = torch.full((H, W), 1e10)
    L = {
        "o_c": rgba(a_obj, (0.8, 0.3, 0.2)), "o_d": torch.where(a_obj > 0, 3.0 + torch.rand(H, W, generator=g), far),
        "s_c": rgba(torch.ones(H, W), (0.7, 0.7, 0.7)), "s_d": 4.0 + torch.rand(H, W, generator=g),
        "o_s_c": rgba(torch.clamp(blob(0.45, 0.62, 0.16), 0, 1), (0.4, 0.4, 0.4)),
        "o_gs_c": rgba(a_gs, (0.2, 0.6, 0.3)), "o_gs_d": torch.where(a_gs > 0, 2.5 + torch.rand(H, W, generator=g), far),
        "s_f_c": rgba(a_smoke, (0.5, 0.5, 0.5)), "s_f_d": 2.0 + 3.0 * torch.rand(H, W, generator=g),
        "s_f_c_pre": rgba(a_smoke * 0.5, (0.9, 0.5, 0.1)),
    }
    L = {k: v.contiguous().to(device) for k, v in L.items()}
    L["s_f_d"] = smoke_depth_fill(L["s_f_c"], L["s_f_d"])
    return L


def timed_regions(fn, regions, distributed, device, local=None):
    """[seconds] of `regions` executions of fn(), each bracketed by barrier + synchronize; max over ranks.  `local`
    (a list) receives this rank's own seconds per region, before the max."""
    import torch.distributed as dist
    out = []
    for _ in range(regions):
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        out.append(time.perf_counter() - t0)
    if local is not None:
        local.extend(out)
    if distributed:
        t = torch.tensor(out, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out = [float(v) for v in t.tolist()]
    return out


def quick(key, device, side, streams, steps, warmup, regions, boundary="op", gaussians=0):
    """A short run of another workload for the `also` object: median-region frames/s, single-stream stage times."""
    from autovfx_amd import _lib
    b = Bench(key, device, None, boundary=boundary, gaussians=gaussians)
    rgba = torch.empty((min(steps, 64), 4, b.H, b.W), dtype=torch.uint8, device=device)
    composed = torch.empty((min(steps, 64), b.H, b.W, 4), dtype=torch.uint8, device=device) if b.layers is not None else None
    frames = [(warmup + j) % b.F for j in range(steps)]
    for f in frames:   # camera matrices resident before the clock starts
        b.cam(f)
    b.run([j % b.F for j in range(warmup)], rgba, streams, side, composed)
    secs = timed_regions(lambda: b.run(frames, rgba, streams, side, composed), regions, False, device)
    med = sorted(secs)[len(secs) // 2]
    _lib.set_stage_timing(True)
    b.run(frames[:12], rgba, 1, side, composed)
    torch.cuda.synchronize()
    st = _lib.stage_times_ms()
    call_ms = sorted(_lib.call_times_ms())
    _lib.set_stage_timing(False)
    st.pop("calls")
    V, D = b.stats(frames[::max(1, steps // 4)])
    out = {"workload": b.name, "boundary": boundary, "value": round(steps / med, 1), "unit": "frames/s", "steps": steps,
           "regions": regions, "streams": streams, "ms_per_step": round(med / steps * 1e3, 4),
           "single_stream_ms_p50": round(call_ms[len(call_ms) // 2], 4) if call_ms else None,
           "stage_ms": {k: round(v, 4) for k, v in st.items()}, "P": b.P, "visible_mean": round(V, 1),
           "num_rendered_mean": round(D, 1), "pairs_per_gaussian": round(D / max(1, b.P), 2)}
    return b, out


# ---------------------------------------------------------------------------------------------------------------------
def launch_command(argv, gpus, port=None, environ=None):
    """What ``python bench.py --gpus N ...`` (N > 1, no WORLD_SIZE in the environment) turns into: the command line and
    environment of a ``torch.distributed.run`` child with one rank per GPU on this node, rendezvous on 127.0.0.1 at a free
    port.  ``argv`` = the arguments bench.py was started with.  HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of these
    boxes only supports dmabuf IPC handles; without it RCCL's peer-to-peer set-up fails with
    ``hipIpcGetMemHandle: invalid argument``."""
    if port is None:
        import socket
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(gpus)}", "--master-addr", "127.0.0.1",
           "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ if environ is None else environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")   # (what torchrun would set, with a warning, when it is unset)
    return cmd, env


def self_launch(args):
    """``--gpus N`` typed without a launcher: start the N ranks ourselves and hand their exit code on.  Rank 0 of the
    child prints the one JSON line to the stdout it inherits from this process."""
    cmd, env = launch_command(sys.argv[1:], args.gpus)
    if os.environ.get("BENCH_LAUNCH_DRY_RUN") == "1":   # tests: show what would run, run nothing
        print(json.dumps({"launch": cmd, "env": {k: env[k] for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "OMP_NUM_THREADS")}}))
        return 0
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("BENCH_LAUNCH_ALLOW_NO_GPU") != "1":   # (the GPU-less launch probe of the tests)
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); one rank per GPU is the only layout "
                         "(RCCL refuses two ranks on one device)")
    sys.stdout.flush()
    return subprocess.call(cmd, env=env)


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HIP runtime starts (see launch_command)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--regions", type=int, default=5, help="how many times the K-step timed region is repeated (median reported)")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--gaussians", type=int, default=0, help="override P (parity/debug only; invalidates the metric)")
    ap.add_argument("--job-frames", type=int, default=0,
                    help="strong scaling: time ONE job of this many frames over all ranks (cloud broadcast from rank 0, "
                         "round-robin shards, pipelined gather) instead of K frames per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-hip", action="store_true",
                    help="skip timing the reference's own kernels compiled for gfx950 (oracle/_ref/libgsr_ref_hip.so)")
    ap.add_argument("--no-also", action="store_true", help="skip the short runs of the other workloads (`also`)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL gather of the frames to rank 0 (N > 1)")
    ap.add_argument("--force-distributed", action="store_true",
                    help="debug: take the N > 1 code path (process group, barriers, pipelined RCCL gather) even with "
                         "one rank, so that path can be exercised on a one-GPU box")
    ap.add_argument("--gather-chunks", type=int, default=16,
                    help="N > 1: pieces the per-rank frame stack is gathered in, each overlapped with the next piece's rendering")
    ap.add_argument("--streams", type=int, default=11,
                    help="render frames on this many HIP streams so independent frames overlap (frame_parallel.DEFAULT_STREAMS: "
                         "the runtime spreads them over 4 hardware queues; same-box A/B in scripts/gpu_ab_flags.sh)")
    ap.add_argument("--driver", choices=["auto", "pipelined", "threads"], default="auto",
                    help="N > 1 path only: how render_and_gather feeds the streams (auto = pipelined)")
    ap.add_argument("--boundary", choices=["op", "render"], default="op",
                    help="op: one GaussianRasterizer.forward per frame (the headline); render: the reference's "
                         "whole per-frame render() = activations + SH pass + normal pass + normal post-processing")
    ap.add_argument("--no-cull", action="store_true", help="A/B knob: GSR_OPT_TILE_CULL = 0")
    ap.add_argument("--slabs", type=int, default=None, help="A/B knob: GSR_OPT_SLABS (1 = no depth slabs; default: as the scene calls for)")
    ap.add_argument("--slab-first", type=int, default=None, help="A/B knob: GSR_OPT_SLAB_FIRST (pairs per tile in the first slab)")
    ap.add_argument("--no-defer-colour", action="store_true", help="A/B knob: GSR_OPT_DEFER_COLOUR = 0")
    ap.add_argument("--frames-digest", action="store_true",
                    help="add `frames_digest` to the line: SHA-256 of the RGBA8 frames rank 0 holds after the last timed region "
                         "(the gathered stack in frame order with N > 1 ranks / --force-distributed), so that the N > 1 code "
                         "path can be checked byte for byte against the N = 1 path (tests/test_rccl_gpu.py)")
    ap.add_argument("--profile-run", action="store_true",
                    help="for rocprofv3: only warm-up + ONE timed region of uniform calls (no replay, statistics, baselines)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner to the C-level stdout when a process
    # group is created (seen with RCCL 2.26), so the real stdout is set aside for that line and everything else
    # that writes to file descriptor 1 lands on stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world   # (under a launcher the launcher's world size is the truth)
    if os.environ.get("BENCH_LAUNCH_PROBE") == "1":
        # The launch path without a GPU (tests/test_bench_launch.py): every rank joins a gloo group at the rendezvous the
        # launcher handed it, the ranks count each other with an all-reduce, rank 0 prints the one line.  Nothing is rendered.
        import torch.distributed as dist
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.ones(1)
        dist.all_reduce(t)
        dist.barrier()
        if rank == 0:
            os.write(json_fd, (json.dumps({"probe": True, "n_ranks_seen": dist.get_world_size(), "ranks_counted": int(t.item()),
                                           "n_gpus": world, "hsa_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}) + "\n").encode())
        dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the render path has no CPU fallback")
    from autovfx_amd.frame_parallel import local_device
    device = local_device()                     # cuda:LOCAL_RANK: one process per GPU
    assert device.index == local_rank
    torch.cuda.set_device(device)

    import torch.distributed as dist
    distributed = world > 1 or args.force_distributed
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    from autovfx_amd import _lib
    from autovfx_amd.frame_parallel import (broadcast_cloud, frames_in_order, rasterize, rasterize_begin, render_and_gather,
                                            shard_frames, side_streams)

    if args.no_cull:
        _lib.set_option(_lib.OPT_TILE_CULL, 0)
    if args.slabs is not None:
        _lib.set_option(_lib.OPT_SLABS, args.slabs)
    if args.slab_first is not None:
        _lib.set_option(_lib.OPT_SLAB_FIRST, args.slab_first)
    if args.no_defer_colour:
        _lib.set_option(_lib.OPT_DEFER_COLOUR, 0)

    b = Bench(args.workload, device, args, boundary=args.boundary, gaussians=args.gaussians)
    W, H, F, P, M, T = b.W, b.H, b.F, b.P, b.M, b.T
    K, Wm, R = args.steps, args.warmup, max(1, args.regions)
    if args.profile_run:
        R = 1
    S = max(1, args.streams)
    side = side_streams(device, S) if S > 1 else []   # the driver's own: allocator pools warmed here stay warm there
    driver = "pipelined" if args.driver == "auto" else args.driver
    strong = args.job_frames > 0
    # rank r renders frames r, r+N, ... ; warm-up frames precede the timed ones on the same orbit
    frame_of = lambda i: (i * world + rank) % F
    if strong:
        my_ids = shard_frames(args.job_frames, rank, world)
        K = (args.job_frames + world - 1) // world   # "steps" = frames of the longest shard
    rgba = torch.empty((max(1, min(K, 256)), 4, H, W), dtype=torch.uint8, device=device)
    composed = torch.empty((rgba.shape[0], H, W, 4), dtype=torch.uint8, device=device) if b.layers is not None else None

    def render_fn_boundary(_cloud, cam, bg_):
        from autovfx_amd import renderer
        out = renderer.render(cam, b.model, renderer.PipelineParams, bg_)
        return out["render"][:3], out["depth"][None], out["render"][3:4], out["radii"]

    class _PendingBoundary:
        def __init__(self, pending):
            self.pending = pending

        def ready(self):
            return self.pending.ready()

        def finish(self):
            out = self.pending.finish()
            return out["render"][:3], out["depth"][None], out["render"][3:4], out["radii"]

    def begin_fn_boundary(_cloud, cam, bg_):
        from autovfx_amd import renderer
        return _PendingBoundary(renderer.render_begin(cam, b.model, renderer.PipelineParams, bg_))

    begin_fn = begin_fn_boundary if args.boundary == "render" else rasterize_begin
    render_fn = render_fn_boundary if args.boundary == "render" else rasterize
    gather_stats, gathered_shape, last_frames = {}, [None], [None]

    timed_frames = [frame_of(Wm + j) for j in range(K)]
    for f in timed_frames:   # camera matrices resident before the clock starts
        b.cam(f)
    if distributed and not args.no_gather and not strong:
        cam_list = [b.cam(f) for f in timed_frames]

        def region():
            g = render_and_gather(b.cloud, cam_list, list(range(K)), b.bg, dst=0, streams=S, chunks=args.gather_chunks,
                                  driver=driver, begin_fn=begin_fn, render_fn=render_fn, rows=K, stats=gather_stats)
            gathered_shape[0] = None if g is None else tuple(g.shape)
            last_frames[0] = None if g is None else (g, K * world)   # (put into frame order after the clock stopped)
    elif strong:
        job_cams = [b.cam(f) for f in range(min(args.job_frames, F))]
        job_cams = [job_cams[f % len(job_cams)] for f in range(args.job_frames)]
        src_cloud = b.cloud

        def region():
            # the whole job: the cloud reaches every rank (rank 0 "has the file"), shards, pipelined gather
            cloud = broadcast_cloud(src_cloud if rank == 0 else None, src=0, device=device) if distributed else src_cloud
            g = render_and_gather(cloud, job_cams, my_ids, b.bg, dst=0, streams=S, chunks=args.gather_chunks, driver=driver,
                                  begin_fn=begin_fn, render_fn=render_fn, rows=K, stats=gather_stats)
            if g is not None:   # rank-major [world, rows, 4, H, W]; frames_in_order (a copy) runs after the clock stopped
                last_frames[0] = (g, args.job_frames)
                gathered_shape[0] = tuple(g.shape)
    else:
        def region():
            b.run(timed_frames, rgba, S, side, composed)
            last_frames[0] = (rgba[:K][None], K) if K <= rgba.shape[0] else None

    with torch.no_grad():
        b.run([frame_of(i) for i in range(Wm)], rgba, S, side, composed)
        if S > 1 and Wm < 2 * S + 2:   # every stream's allocator pool and scratch sizes settle before the clock starts
            b.run(timed_frames[:2 * S + 2 - Wm], rgba, S, side, composed)
        if distributed or strong:
            region()   # the gather path's own rehearsal: frame stack, receive buffers, RCCL's first gather (untimed)
        _lib.set_stage_timing(False)
        local_secs = []
        secs = timed_regions(region, R, distributed, device, local=local_secs)
    # what RCCL saw: the group's size and every rank's own rate over its own median region (all-gathered)
    n_ranks_seen, per_rank_fps = 1, None
    my_frames = len(my_ids) if strong else K
    my_fps = my_frames / sorted(local_secs)[len(local_secs) // 2]
    if distributed:
        n_ranks_seen = dist.get_world_size()
        t = torch.zeros(n_ranks_seen, dtype=torch.float64, device=device)
        t[rank] = my_fps
        dist.all_reduce(t)
        per_rank_fps = [round(float(v), 2) for v in t.tolist()]
    else:
        per_rank_fps = [round(my_fps, 2)]
    # N x single-GPU, measured in THIS run: rank 0 renders the same frames alone -- no gather, the other ranks waiting at a
    # barrier -- so that the line carries its own yardstick (the driver computes efficiency from separate N = 1 runs as well)
    alone_fps = None
    if distributed and not strong and not args.profile_run:
        with torch.no_grad():
            dist.barrier()
            if rank == 0:
                b.run(timed_frames[:min(K, 2 * S + 2)], rgba, S, side, composed)
                asecs = timed_regions(lambda: b.run(timed_frames, rgba, S, side, composed), min(R, 3), False, device)
                alone_fps = K / sorted(asecs)[len(asecs) // 2]
            dist.barrier()

    if args.profile_run:
        if rank == 0:
            os.write(json_fd, (json.dumps({"profile_run": True, "steps": K, "warmup": Wm, "streams": S,
                                           "ms_per_step": round(secs[0] / K * 1e3, 4)}) + "\n").encode())
        if distributed:
            dist.destroy_process_group()
        return

    # ---- stage breakdown: an untimed single-stream replay of the first timed frames, HIP events on the launch stream ----
    with torch.no_grad():
        _lib.set_stage_timing(True)
        b.run(timed_frames[:min(K, 32)], rgba, 1, side, composed)
        torch.cuda.synchronize()
        stage_ms = _lib.stage_times_ms()
        call_ms = sorted(_lib.call_times_ms())
        slab_pairs = _lib.slab_pairs()
        _lib.set_stage_timing(False)

    order = sorted(secs)
    med = order[len(order) // 2]
    total_frames = args.job_frames if strong else K * world
    fps = lambda s: total_frames / s
    ms_per_step = med / K * 1e3

    # ---- the caller's own loop: one blocking GaussianRasterizer.forward (or render()) per frame, one stream ----
    serial = None
    if world == 1 and not distributed and not strong and b.extra is None and b.layers is None:
        Ks = min(K, 100)
        with torch.no_grad():
            b.run_blocking(timed_frames[:min(Ks, 8)], rgba)
            ssecs = timed_regions(lambda: b.run_blocking(timed_frames[:Ks], rgba), min(R, 3), False, device)
        smed = sorted(ssecs)[len(ssecs) // 2]
        serial = {"value": round(Ks / smed, 2), "unit": "frames/s", "ms_per_step": round(smed / Ks * 1e3, 4), "steps": Ks,
                  "regions": [round(Ks / x, 2) for x in ssecs], "streams": 1,
                  "what": "what an unmodified caller gets (scene_representation.py:355-424): frames one at a time on the "
                          "current stream, each a blocking " + ("render()" if args.boundary == "render" else "GaussianRasterizer.forward")
                          + " + RGBA8 pack; host wall clock between synchronize()s, like `value`"}

    # ---- per-frame workload statistics (untimed): the units of this rank's timed frames ----
    units = b.units(timed_frames[::max(1, K // 8)])
    V, D = units["V"], units["D"]
    alg = algorithmic_bytes(P, V, D, T, W, H, M)
    design = design_bytes(units, inference=True, deferred=_lib.get_option(_lib.OPT_DEFER_COLOUR) != 0)

    calls = stage_ms.pop("calls")
    traffic = pmc_traffic(T_key=args.workload if not args.gaussians else None)
    counter = stage_counter_bytes(traffic)
    # Per stage: this library's design bytes (design_bytes above: what its own layout has to move, measured units) and the
    # HBM bytes the counters saw (profiles/, per stage) over the stage's time.  Both are <= what the hardware can move; the
    # reference formula's bytes are NOT credited per stage -- the stages are not the reference's -- only
    # for the frame (`frac_equiv`) and for the dominant kernel, as the contract asks.
    stages = {}
    for k, v in stage_ms.items():
        ent = {"ms": round(v, 4), "design_bytes": int(design.get(k, 0)),
               "design_GBps": round(design.get(k, 0) / (v * 1e-3) / 1e9, 1) if v > 0 else None}
        if counter is not None and k in counter:
            ent["counter_bytes"] = int(counter[k])
            ent["counter_GBps"] = round(counter[k] / (v * 1e-3) / 1e9, 1) if v > 0 else None
            ent["frac_of_hbm_peak"] = round(counter[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if v > 0 else None
        stages[k] = ent
    # the dominant kernel of the frame is the blend; an inference call launches it once per depth slab, so "per launch"
    # figures are the frame's blend figures divided by the launches of a frame (what rocprofv3's per-kernel average is)
    stage_alg = {"preprocess": alg["preprocess"], "depth_sort": 0, "scan": alg["scan"], "duplicate": alg["duplicate"],
                 "tile_sort": alg["sort"], "ranges": alg["ranges"], "blend": alg["blend"], "colour": 0}
    dom = max(stage_ms, key=stage_ms.get)
    kernel_of = {"blend": "blend_quadrant_kernel", "preprocess": "preprocess_kernel", "duplicate": "expand_kernel",
                 "ranges": "tile_ranges_kernel", "colour": "sh_colour_listed_kernel"}
    launches = max(1, len(slab_pairs)) if dom in ("blend", "duplicate", "ranges", "colour", "tile_sort") else 1
    dom_kernel = kernel_of.get(dom, dom)
    dom_traffic = None
    if traffic is not None and dom_kernel in traffic["kernels"]:
        dom_traffic = int(traffic["kernels"][dom_kernel]["bytes_per_launch"])
    dom_bytes = stage_alg[dom] if stage_alg[dom] else design[dom]
    dom_gbps = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    equiv_gbps = alg["frame"] / (ms_per_step * 1e-3) / 1e9
    design_gbps = design["frame"] / (ms_per_step * 1e-3) / 1e9
    frame = {"ms_per_step": round(ms_per_step, 4),
             # one frame alone on the GPU, first kernel to last (HIP events), over the per-stage replay
             "single_stream_ms_p50": round(call_ms[len(call_ms) // 2], 4) if call_ms else None,
             "single_stream_ms_p95": round(call_ms[min(len(call_ms) - 1, int(0.95 * len(call_ms)))], 4) if call_ms else None,
             "design_bytes": int(design["frame"]), "achieved_design": round(design_gbps, 1),
             "frac_design": round(design_gbps / HBM_PEAK_GBPS, 4),
             # the reference formula's bytes (SURVEY.md 8d B_alg, 6 radix passes over 64-bit keys ...) over this frame time:
             # an EQUIVALENCE figure -- this design does not move most of those bytes
             "alg_bytes_reference_formula": int(alg["frame"]), "achieved_equiv": round(equiv_gbps, 1),
             "frac_equiv": round(equiv_gbps / HBM_PEAK_GBPS, 4), "n_pass_ref_sort": alg["n_pass"]}
    if traffic is not None:
        tr_gbps = traffic["frame_bytes"] / (ms_per_step * 1e-3) / 1e9
        # THE frame figure: HBM bytes the counters measured for one frame / the frame time of the timed regions / peak
        frame.update({"traffic": int(traffic["frame_bytes"]), "achieved": round(tr_gbps, 1),
                      "frac": round(tr_gbps / HBM_PEAK_GBPS, 4)})
        if serial is not None:
            frame["frac_serial"] = round(traffic["frame_bytes"] / (serial["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    else:
        frame.update({"traffic": None, "achieved": round(design_gbps, 1), "frac": round(design_gbps / HBM_PEAK_GBPS, 4),
                      "frac_is": "design bytes (no counter summary for this workload)"})
    roofline = {
        "bound": "hbm", "kernel": dom_kernel,
        "achieved": round(dom_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
        "frac": round(dom_gbps / HBM_PEAK_GBPS, 4), "traffic": dom_traffic,
        "traffic_source": None if traffic is None else traffic["source"],
        "traffic_measured_at": None if traffic is None else traffic["commit"],
        "launches_per_frame": launches, "avg_launch_ms": round(stage_ms[dom] / launches, 4),
        "alg_bytes_per_launch": int(dom_bytes / launches), "per_frame_ms": round(stage_ms[dom], 4),
        "alg_bytes_per_frame": int(dom_bytes), "design_bytes_per_frame": int(design[dom]), "stage_replay_calls": calls,
        "note": "vector-instruction-issue bound, not HBM bound (DESIGN.md section 4): frac = SURVEY 8d's algorithmic bytes of the "
                "kernel (44 B per reference pair + 24 B per pixel) over its time and the HBM peak",
        "frame": frame, "stages": stages, "slab_pairs_last_frame": slab_pairs,
        "units_mean": {k: (round(v, 1) if not isinstance(v, list) else [round(x, 1) for x in v]) for k, v in units.items()},
    }

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.boundary == "op" and b.key in ("c3", "c2", "heavy", "heavy1080"):
        cpu_baseline = run_cpu_baseline(b)
    reference_on_gpu = None
    if rank == 0 and world == 1 and args.boundary == "op" and not args.no_reference_hip and b.extra is None and b.layers is None:
        reference_on_gpu = run_reference_on_gpu(b.cloud, [b.cam(f) for f in timed_frames[:24]], b.bg)

    also = None
    if rank == 0 and world == 1 and not args.no_also and args.workload == "c3" and not args.gaussians and args.boundary == "op":
        also = run_also(device, side, S, parity=not args.no_cpu_baseline)

    if rank == 0:
        metric = ("rendered frames/sec at 1920x1080, 3M Gaussians" if args.workload == "c3" and not args.gaussians
                  else f"rendered frames/sec at {W}x{H}, {P} Gaussians ({args.workload})")
        line = {
            "metric": metric + (" [render() boundary]" if args.boundary == "render" else ""),
            "value": round(fps(med), 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "regions": {"count": R, "steps_each": K, "value_median": round(fps(med), 2), "value_min": round(fps(order[-1]), 2),
                        "value_max": round(fps(order[0]), 2), "values": [round(fps(s), 2) for s in secs]},
            "config": {"workload": b.name, "P": P, "sh_coeffs": M, "width": W, "height": H, "tiles": T,
                       "visible_mean": round(V, 1), "num_rendered_mean": round(D, 1),
                       "boundary": ("render(): activations, SH pass, normal pass (folded into the first), normal/pseudo-normal "
                                    "post-processing, RGBA8 pack per frame" if args.boundary == "render" else
                                    "GaussianRasterizer.forward + RGBA8 pack per frame")
                                   + ("; composite with 5 Blender layers per frame" if b.layers is not None else "")
                                   + ("; RCCL gather of the RGBA8 frames to rank 0, pipelined behind the rendering"
                                      if distributed and not args.no_gather else "")
                                   + ("; ONE job: cloud broadcast from rank 0 + round-robin shards + gather, all timed" if strong else ""),
                       "parallelism": f"frame-parallel x{world}", "streams_per_gpu": S,
                       "stream_driver": ("pipelined" if not (distributed or strong) else driver) if S > 1 else "serial",
                       "options": {"tile_cull": _lib.get_option(_lib.OPT_TILE_CULL),
                                   "slabs": _lib.get_option(_lib.OPT_SLABS), "slab_first": _lib.get_option(_lib.OPT_SLAB_FIRST),
                                   "defer_colour": _lib.get_option(_lib.OPT_DEFER_COLOUR),
                                   "depth_drop": _lib.get_option(_lib.OPT_DEPTH_DROP), "blend_order": _lib.get_option(_lib.OPT_BLEND_ORDER),
                                   "radix_rank_active": _lib.get_option(_lib.OPT_RADIX_RANK_ACTIVE),
                                   # 4096-key tiles of this whole run whose LDS-add ranks failed the sort's own order check
                                   "radix_rank_fallbacks": _lib.get_option(_lib.OPT_RADIX_RANK_FALLBACKS)}},
            "n_ranks_seen": n_ranks_seen, "per_rank_frames_per_s": per_rank_fps,
            "value_serial": None if serial is None else serial["value"],
            "ms_per_step_serial": None if serial is None else serial["ms_per_step"],
            "serial": serial, "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        if strong:
            line["config"]["job_frames"] = args.job_frames
        if gather_stats:
            line["config"]["gather"] = {"chunks": args.gather_chunks, "gathered_shape": gathered_shape[0],
                                        "last_region_render_s": round(gather_stats.get("render_s", 0.0), 5),
                                        "last_region_gather_tail_ms": round(gather_stats.get("gather_tail_s", 0.0) * 1e3, 3),
                                        "per_rank_frames_per_s_rank0": round(K / max(1e-9, gather_stats.get("render_s", 0.0)), 1)}
        if reference_on_gpu is not None:
            line["reference_on_gpu"] = reference_on_gpu
        if also is not None:
            line["also"] = also
        # scalars a reader of the driver's record needs, where that record keeps them (its copy of `roofline` and `config`
        # holds scalar fields only; the nested objects above stay for the full line)
        rf = line["roofline"]
        rf["frame_frac_traffic"] = frame.get("frac") if traffic is not None else None
        rf["frame_traffic_bytes"] = frame.get("traffic")
        rf["frame_frac_design"] = frame["frac_design"]
        rf["frame_frac_serial"] = frame.get("frac_serial")
        rf["serial_ms_per_step"] = None if serial is None else serial["ms_per_step"]
        rf["serial_value"] = None if serial is None else serial["value"]
        rf["reference_on_gpu_ms"] = reference_on_gpu.get("ms_per_frame") if isinstance(reference_on_gpu, dict) else None
        line["config"]["serial_frames_per_s"] = None if serial is None else serial["value"]
        line["config"]["n_ranks_seen"] = n_ranks_seen
        # the multi-GPU scalars where the driver's record keeps scalars: every rank's own rate (min / max; the list is `per_rank_frames_per_s`),
        # the gather's tail behind the last rendered frame, and the whole job against N times rank 0 alone in this same run
        line["config"]["per_rank_frames_per_s_min"] = min(per_rank_fps) if per_rank_fps else None
        line["config"]["per_rank_frames_per_s_max"] = max(per_rank_fps) if per_rank_fps else None
        line["config"]["gather_tail_ms"] = round(gather_stats.get("gather_tail_s", 0.0) * 1e3, 3) if gather_stats else None
        line["config"]["gather_render_s"] = round(gather_stats.get("render_s", 0.0), 5) if gather_stats else None
        line["config"]["single_gpu_frames_per_s_same_run"] = None if alone_fps is None else round(alone_fps, 2)
        line["config"]["frac_of_n_x_single_gpu"] = None if alone_fps is None else round(fps(med) / (world * alone_fps), 4)
        if isinstance(also, dict):
            ur = also.get("c3_reference_shaped_render") or {}
            bw = also.get("backward_c3") or {}
            rf["unchanged_render_ms_per_frame"] = ur.get("ms_per_frame")
            rf["unchanged_render_pytorch_prep_ms_per_frame"] = ur.get("pytorch_prep_ms_per_frame")
            rf["backward_kernel_ms"] = (bw.get("kernels") or {}).get("render_backward_kernel", {}).get("ms")
            rf["training_iteration_ms"] = bw.get("ms_per_iter")
            rf["training_iteration_reference_on_gpu_ms"] = (bw.get("reference_on_gpu") or {}).get("ms_per_iter")
            tr = also.get("training_render_c3") or {}
            rf["training_render_iteration_ms"] = tr.get("ms_per_iter")
            rf["training_render_iteration_reference_structure_ms"] = tr.get("reference_structure_ms_per_iter")
            line["config"]["unchanged_render_frames_per_s"] = ur.get("value")
            # ... and the two whole functions of the unchanged process beside the op-level headline (C2-sized, to a tmpfs): the frame
            # loop with its four files per frame, and blend_frames with its ~11 layers per frame
            lp, bl = also.get("c5_loop") or {}, also.get("c5_blend_frames") or {}
            line["config"]["frame_loop_c2_frames_per_s"] = lp.get("value")
            line["config"]["frame_loop_c2_reference_shaped_frames_per_s"] = (lp.get("reference_shaped_loop") or {}).get("frames_per_s")
            line["config"]["blend_frames_frames_per_s"] = bl.get("value")
            line["config"]["blend_frames_reference_shaped_frames_per_s"] = (bl.get("reference_shaped") or {}).get("frames_per_s")
        if args.frames_digest and last_frames[0] is not None:
            import hashlib
            torch.cuda.synchronize()
            fr = frames_in_order(*last_frames[0]).contiguous().cpu().numpy()
            line["frames_digest"] = {"sha256": hashlib.sha256(fr.tobytes()).hexdigest(), "shape": list(fr.shape),
                                     "nonzero_bytes": int(np.count_nonzero(fr))}
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


def run_also(device, side, S, parity=True):
    """Short runs of the other BASELINE configs and boundaries, so that their figures are on the driver-observed line."""
    also = {}

    def guarded(name, fn):
        try:
            also[name] = fn()
        except Exception as e:   # an extra must not break the headline line
            also[name] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()

    guarded("c2_op", lambda: quick("c2", device, side, S, 200, 20, 3)[1])
    guarded("c3_render_boundary", lambda: quick("c3", device, side, S, 60, 10, 3, boundary="render")[1])

    def heavy(key):
        b, out = quick(key, device, side, S, 60, 8, 3)
        ref = run_reference_on_gpu(b.cloud, [b.cam(f) for f in range(8, 20)], b.bg)
        if ref is not None:
            out["reference_on_gpu"] = ref
            if "value" in ref:
                out["vs_reference_kernels"] = round(out["value"] / ref["value"], 2)
        return out

    guarded("heavy_op", lambda: heavy("heavy"))
    guarded("heavy1080_op", lambda: heavy("heavy1080"))
    guarded("c4_fused_rgb_normal_depth", lambda: quick("c4", device, side, S, 50, 10, 3)[1])

    def c5():
        from autovfx_amd.compositor import composite_frame
        b, out = quick("c5", device, side, S, 400, 20, 3)
        # the compositor alone: 52 B per pixel with every layer present (4 B x 6 colour layers + 4 B x 4 depth maps in,
        # ... 4 B out), HIP events around 200 launches
        L = b.layers
        bg_c = torch.zeros((b.H, b.W, 4), dtype=torch.uint8, device=device)
        dst = torch.empty_like(bg_c)
        args = (bg_c, L["o_c"], L["o_d"], L["s_c"], L["s_d"], L["o_s_c"], L["o_gs_c"], L["o_gs_d"], L["s_f_c"], L["s_f_d"], L["s_f_c_pre"])
        for _ in range(10):
            composite_frame(*args, out=dst)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            composite_frame(*args, out=dst)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 200
        nbytes = 52 * b.W * b.H
        out["compositor"] = {"avg_launch_ms": round(ms, 5), "alg_bytes_per_launch": nbytes,
                             "achieved_GBps": round(nbytes / (ms * 1e-3) / 1e9, 1),
                             "frac_of_hbm_peak": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
        # the same composite from layers as Blender renders them -- 2x the frame, the reference's anti-aliasing set-up -- resized on
        # the GPU exactly as blend_all.py:217-234 does with PIL (gsr_resize_rgba8_bilinear / gsr_resize_f32_nearest: Pillow's bytes)
        g2 = torch.Generator(device=device).manual_seed(9)
        big_c = lambda: torch.randint(0, 256, (2 * b.H, 2 * b.W, 4), dtype=torch.uint8, device=device, generator=g2)
        big_d = lambda: torch.rand((2 * b.H, 2 * b.W), device=device, generator=g2) * 5 + 1
        args2 = (bg_c, big_c(), big_d(), big_c(), big_d(), big_c(), big_c(), big_d(), big_c(), big_d(), big_c())
        for _ in range(5):
            composite_frame(*args2, out=dst)
        e0.record()
        for _ in range(100):
            composite_frame(*args2, out=dst)
        e1.record()
        torch.cuda.synchronize()
        ms2 = e0.elapsed_time(e1) / 100
        in_bytes = (6 * 4 + 4 * 4) * 4 * b.W * b.H + 8 * b.W * b.H     # ten layers at 2x in (6 RGBA8, 4 fp32), the frame's background in, the frame out
        out["compositor_from_2x_layers"] = {"avg_ms": round(ms2, 5), "alg_bytes": int(in_bytes), "achieved_GBps": round(in_bytes / (ms2 * 1e-3) / 1e9, 1),
                                            "frac_of_hbm_peak": round(in_bytes / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                            "what": "ten Blender layers at 2x the frame resized like PIL (bilinear RGBA8 with the premultiply round trip, "
                                                    "nearest fp32), then the composite: 11 resize launches + 1"}
        out["disk_io"] = time_frame_files(b, 8)
        return out

    guarded("c5_render_and_composite", c5)
    guarded("c5_dynamic", lambda: dynamic_scene_bench(device, side, S))
    guarded("c5_loop", lambda: frame_loop_bench(device))
    guarded("c5_blend_frames", lambda: blend_frames_bench(device))
    guarded("c3_reference_shaped_render", lambda: reference_shaped_render(device))
    guarded("backward_c3", lambda: backward_iteration("c3", device, parity=parity))
    guarded("training_render_c3", lambda: training_render_iteration(device))
    return also


def dynamic_scene_bench(device, side, S, frames=120):
    """BASELINE configs[4] with moving objects (scene_representation.py:357-372): the C2 scene (1 M Gaussians, 960x540) plus
    two inserted objects of 60 k Gaussians, each with its own rigid transform per frame.  Three ways: the static scene (no
    objects move: the buffers are composed once) as the yardstick; DynamicScene (one gsr_place_object launch per object and
    frame into resident buffers); the reference's structure in PyTorch on the GPU (clone the scene, transform with ~10
    launches per object, six concatenations over everything, re-activate everything) -- WITHOUT its per-frame PLY reload."""
    import math
    from autovfx_amd import scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.dynamic_scene import DynamicScene
    from oracle.dynamic_torch import reference_shaped_compose
    from autovfx_amd.frame_parallel import pack_rgba8, rasterize, rasterize_begin
    from autovfx_amd.gaussian_model import GaussianModel
    W, H = 960, 540
    c = scenes.config_c2()
    base = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3)
    objs = {}
    for k, name in enumerate(("a", "b")):
        o = scenes.config_c1(P=60_000, seed=70 + k)
        objs[name] = (GaussianModel.from_activated(o.means3D * 0.25, o.opacities, o.scales * 0.25, o.rotations, o.shs, 3), (0.0, 0.0, 0.0))

    def rz(deg):
        t = math.radians(deg)
        return np.array([[math.cos(t), -math.sin(t), 0], [math.sin(t), math.cos(t), 0], [0, 0, 1]], np.float32)

    place = lambda f: [("a", (0.8 * math.cos(0.05 * f), 0.8 * math.sin(0.05 * f), 0.2), rz(3.0 * f), 1.0),
                       ("b", (-0.5, 0.3 + 0.004 * f, 0.1 * math.sin(0.1 * f)), rz(-2.0 * f), 1.0 + 0.002 * f)]
    cams = [cam.to(device) for cam in orbit_cameras(200, W, H)[:frames]]
    bg = torch.zeros(3, device=device)
    rgba = torch.empty((4, H, W), dtype=torch.uint8, device=device)
    scene = DynamicScene(base, objs, device=device, slots=max(1, S))
    static_cloud = scene.compose(place(0))

    def serial(compose):
        with torch.no_grad():
            for f in range(frames):
                color, _d, alpha, _r = rasterize(compose(f), cams[f], bg)
                pack_rgba8(color, alpha, out=rgba)

    def timed(fn):
        fn()
        secs = timed_regions(fn, 3, False, device)
        return sorted(secs)[1] / frames * 1e3

    out = {"workload": "C2 scene (1 M Gaussians, 960x540) + 2 inserted objects of 60 k Gaussians, each moved rigidly every frame",
           "frames": frames, "P_frame": static_cloud.P,
           "sh_degree_of_frames_with_objects": static_cloud.sh_degree,   # 0: what the reference's merged model renders at (round 4)
           }
    out["static_ms_per_frame"] = round(timed(lambda: serial(lambda f: static_cloud)), 4)
    out["ms_per_frame"] = round(timed(lambda: serial(lambda f: scene.compose(place(f)))), 4)
    out["reference_shaped_ms_per_frame"] = round(timed(lambda: serial(lambda f: reference_shaped_compose(base, objs, place(f), device))), 4)
    # the melting branch (scene_representation.py:373-421): per frame a different masked subset of each object is merged
    # untransformed (gsr_place_object_subset); the masks are resident index lists, as a caller that matched meshes once would hold
    g = torch.Generator(device="cpu").manual_seed(3)
    n_obj = {k: int(m._xyz.shape[0]) for k, (m, _c) in objs.items()}
    subsets = [{k: torch.nonzero(torch.rand(n, generator=g) < (0.25 + 0.5 * ((f * 7 + j) % 5) / 4)).reshape(-1).to(torch.int32).to(device)
                for j, (k, n) in enumerate(n_obj.items())} for f in range(8)]
    masked = lambda f: [(k, None, None, None, subsets[f % 8][k]) for k in n_obj]
    static_masked = scene.compose(masked(0))
    out["masked_static_ms_per_frame"] = round(timed(lambda: serial(lambda f: static_masked)), 4)
    out["masked_ms_per_frame"] = round(timed(lambda: serial(lambda f: scene.compose(masked(f)))), 4)
    out["masked_vs_static"] = round(out["masked_static_ms_per_frame"] / out["masked_ms_per_frame"], 3)
    out["value"] = round(1e3 / out["ms_per_frame"], 1)
    out["unit"] = "frames/s"
    out["vs_static"] = round(out["static_ms_per_frame"] / out["ms_per_frame"], 3)
    out["streams"] = 1
    # several frames in flight: frame f composed into slot f % S on the stream that renders it
    if S > 1:
        from collections import deque

        def pipelined():
            with torch.no_grad():
                q = deque()

                def finish():
                    st, p = q.popleft()
                    with torch.cuda.stream(st):
                        color, _d, alpha, _r = p.finish()
                        pack_rgba8(color, alpha, out=rgba)

                for f in range(frames):
                    while len(q) == S:
                        finish()
                    st = side[f % S]
                    with torch.cuda.stream(st):
                        q.append((st, rasterize_begin(scene.compose(place(f), slot=f % S), cams[f], bg)))
                while q:
                    finish()
                for st in side:
                    torch.cuda.current_stream(device).wait_stream(st)

        ms = timed(pipelined)
        out["pipelined"] = {"streams": S, "ms_per_frame": round(ms, 4), "value": round(1e3 / ms, 1)}
    with torch.no_grad():   # parity of one moving frame against the reference-shaped composition on the same GPU
        a = rasterize(scene.compose(place(37)), cams[37], bg)[0].clone()
        b = rasterize(reference_shaped_compose(base, objs, place(37), device), cams[37], bg)[0]
    out["rgb_maxabs_vs_reference_shaped_frame37"] = float((a - b).abs().max())
    return out


# ---------------------------------------------------------------------------------------------------------------------
# also.c5_loop: SceneRepresentation.render_from_3DGS as a whole (scene_representation.py:337-447) -- rebuild of the frame's
# Gaussian set, render(), the four files of every frame on a tmpfs -- through the drop-in and in the reference's shape
# ---------------------------------------------------------------------------------------------------------------------
_LOOP_CENTRES = {}


def load_gaussians(path, max_sh_degree=4):            # what frame_loop takes from "the module of the scene class": here, this file
    from autovfx_amd.gaussian_model import GaussianModel
    return GaussianModel(max_sh_degree).load_ply(path, device="cuda:%d" % torch.cuda.current_device())


def get_center_of_mesh_2(mesh_path):
    return np.asarray(_LOOP_CENTRES[mesh_path], np.float64)


class _LoopScene:
    """The attributes render_from_3DGS reads from the reference's SceneRepresentation (scene_representation.py:47-113,192-221)."""

    def __init__(self, root, results, ply, views, objects, rb_info, device):
        import types
        from autovfx_amd import renderer
        self.hparams = types.SimpleNamespace(max_sh_degree=4, render_type="MULTI_VIEW", blender_output_dir_name="blend")
        self.traj_results_dir = os.path.join(root, results)
        self.blender_cache_dir = os.path.join(root, "no_cache")
        self.cameras = {"cameras": views}
        self.anchor_frame_idx, self.total_frames = 0, len(views)
        self.rb_transform_info, self.blender_cfg = rb_info, {"insert_object_info": objects}
        self.background = torch.zeros(3, device=device)
        self.pipe, self._ply, self._device, self.load_seconds = renderer.PipelineParams, ply, device, 0.0

    def load_scene(self):
        from autovfx_amd.gaussian_model import GaussianModel
        t0 = time.perf_counter()
        self.gaussians = GaussianModel(3).load_ply(self._ply, device=str(self._device))
        torch.cuda.synchronize()
        self.load_seconds = time.perf_counter() - t0


def _reference_shaped_loop(scene, frame_ids):
    """scene_representation.py:355-438 in the reference's shape on this GPU: per frame a deep copy of the scene, every placed object's PLY
    from disk, transform_gaussians + merge_two_gaussians as PyTorch operations (oracle/dynamic_torch.py), ONE blocking render() -- this
    package's, the unchanged caller's 1 ms -- then the host writers: ``save_image`` (PIL, its default compression), ``.cpu().numpy()``,
    ``np.save``, two ``cv2.imwrite``-equivalents (PNG compression 1, OpenCV's default)."""
    import copy
    from PIL import Image
    from autovfx_amd import renderer
    from autovfx_amd.dynamic_scene import FrameModel
    from autovfx_amd.frame_io import depth2img
    from autovfx_amd.gaussian_model import get_minimum_axis
    from oracle.dynamic_torch import reference_shaped_compose
    out = scene.traj_results_dir
    for sub in ("images", "depth", "normal"):
        os.makedirs(os.path.join(out, sub), exist_ok=True)
    info = {o["object_id"]: o for o in scene.blender_cfg["insert_object_info"]}
    with torch.no_grad():
        for idx in frame_ids:
            view = scene.cameras["cameras"][idx]
            base = copy.deepcopy(scene.gaussians)
            key, placed, objs = "{0:03d}".format(idx + 1), [], {}
            for oid, t in scene.rb_transform_info.items():
                if key in t:
                    path = os.path.join("/".join(info[oid]["object_path"].split("/")[:-2]), "object_gaussians.ply")
                    objs[oid] = (load_gaussians(path, 3), get_center_of_mesh_2(info[oid]["object_path"]))
                    placed.append((oid, t[key]["pos"], t[key]["rot"], t[key]["scale"]))
            cloud = reference_shaped_compose(base, objs, placed, scene._device)
            model = FrameModel(cloud, get_minimum_axis(cloud.scales, cloud.rotations).contiguous())
            res = renderer.render(view, model, scene.pipe, scene.background)
            rgba = res["render"].mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
            Image.fromarray(rgba).save(os.path.join(out, "images", view.image_name + ".png"))
            depth = res["depth"].cpu().numpy().squeeze()
            np.save(os.path.join(out, "depth", view.image_name + ".npy"), depth.astype(np.float32))
            Image.fromarray(depth2img(depth, 3.0)).save(os.path.join(out, "depth", view.image_name + ".png"), compress_level=1)
            normal = ((res["normal"].cpu().numpy() + 1) / 2 * 255).astype(np.uint8)
            Image.fromarray(normal).save(os.path.join(out, "normal", view.image_name + ".png"), compress_level=1)


def frame_loop_bench(device, frames=400, reference_frames=10):
    """C2 (1 M Gaussians, 960x540) + two inserted objects of 60 k Gaussians that move rigidly every frame, 400 frames (configs[4]):
    ``autovfx_amd.frame_loop.render_from_3DGS`` -- what ``install()`` puts behind ``SceneRepresentation.render_from_3DGS`` -- end
    to end on a tmpfs, and the reference-shaped loop beside it on a bounded number of frames."""
    import math
    import shutil
    import tempfile
    from autovfx_amd import frame_loop, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.gaussian_model import GaussianModel
    W, H = 960, 540
    root = tempfile.mkdtemp(prefix="gsr_loop_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        c = scenes.config_c2()
        ply = os.path.join(root, "scene", "point_cloud.ply")
        GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3).save_ply(ply)
        objects = []
        for k, name in enumerate(("a", "b")):
            o = scenes.config_c1(P=60_000, seed=70 + k)
            d = os.path.join(root, "assets", name)
            GaussianModel.from_activated(o.means3D * 0.25, o.opacities, o.scales * 0.25, o.rotations, o.shs, 3).save_ply(
                os.path.join(d, "object_gaussians.ply"))
            mesh = os.path.join(d, "mesh", name + ".obj")
            _LOOP_CENTRES[mesh] = (0.0, 0.0, 0.0)
            objects.append({"object_id": name, "object_path": mesh})
        rz = lambda deg: [[math.cos(math.radians(deg)), -math.sin(math.radians(deg)), 0.0],
                          [math.sin(math.radians(deg)), math.cos(math.radians(deg)), 0.0], [0.0, 0.0, 1.0]]
        rb = {"a": {}, "b": {}}
        for f in range(frames):
            key = "{0:03d}".format(f + 1)
            rb["a"][key] = {"pos": [0.8 * math.cos(0.05 * f), 0.8 * math.sin(0.05 * f), 0.2], "rot": rz(3.0 * f), "scale": 1.0}
            rb["b"][key] = {"pos": [-0.5, 0.3 + 0.004 * f, 0.1 * math.sin(0.1 * f)], "rot": rz(-2.0 * f), "scale": 1.0 + 0.002 * f}
        cams = orbit_cameras(200, W, H)
        views = [cams[f % 200].to(device) for f in range(frames)]
        for f, v in enumerate(views):
            v.image_name = "{0:05d}".format(f)
        ours = _LoopScene(root, "ours", ply, views, objects, rb, device)
        os.environ["AUTOVFX_AMD_LOOP_STATS"] = "1"                # (the loop's own split of its host time: four clock reads per frame)
        frame_loop.render_from_3DGS(ours)                         # warm: allocator pools, pinned slots, page cache of the PLYs
        shutil.rmtree(ours.traj_results_dir)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frame_loop.render_from_3DGS(ours)
        torch.cuda.synchronize()
        t_call = time.perf_counter() - t0
        t_loop = t_call - ours.load_seconds
        stats_now = dict(frame_loop.LAST_LOOP_STATS)
        files = [os.path.join(ours.traj_results_dir, sub, "00007" + ext) for sub, ext in
                 (("images", ".png"), ("depth", ".npy"), ("depth", ".png"), ("normal", ".png"))]
        nbytes = sum(os.path.getsize(p) for p in files if os.path.exists(p))
        n_written = sum(len(os.listdir(os.path.join(ours.traj_results_dir, sub))) for sub in ("images", "depth", "normal"))
        shutil.rmtree(ours.traj_results_dir, ignore_errors=True)
        host_profile = None
        if os.environ.get("GSR_LOOP_PROFILE"):       # where the host thread's time goes (cProfile slows the loop: not the timed call)
            import cProfile
            import io
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            frame_loop.render_from_3DGS(ours)
            pr.disable()
            shutil.rmtree(ours.traj_results_dir)
            buf = io.StringIO()
            pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(28)
            host_profile = [ln.rstrip() for ln in buf.getvalue().splitlines() if ln.strip()][:45]
        theirs = _LoopScene(root, "theirs", ply, views, objects, rb, device)
        theirs.load_scene()
        ids = list(range(0, frames, max(1, frames // reference_frames)))[:reference_frames]
        _reference_shaped_loop(theirs, ids[:2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _reference_shaped_loop(theirs, ids)
        torch.cuda.synchronize()
        t_ref = time.perf_counter() - t0
        ref_bytes = sum(os.path.getsize(os.path.join(theirs.traj_results_dir, sub, views[ids[0]].image_name + ext)) for sub, ext in
                        (("images", ".png"), ("depth", ".npy"), ("depth", ".png"), ("normal", ".png")))
        return {"workload": "C2 scene (1 M Gaussians, 960x540) + 2 inserted objects of 60 k Gaussians moved rigidly every frame; "
                            "4 files per frame to a tmpfs", "frames": frames, "files_written": n_written,
                "value": round(frames / t_loop, 1), "unit": "frames/s", "ms_per_frame": round(t_loop / frames * 1e3, 4),
                "call_seconds": round(t_call, 3), "load_scene_seconds": round(ours.load_seconds, 3),
                "frames_per_s_including_load_scene": round(frames / t_call, 1),
                # `value` counts everything render_from_3DGS does except re-reading the scene's own PLY (the reference re-reads it too):
                # the objects' PLYs, the resident scene buffers (once per call), the loop, the last file on disk.  The loop alone:
                "frames_per_s_loop_only": (round(frames / stats_now["loop_s"], 1) if stats_now.get("loop_s") else None),
                "per_call_setup_seconds": (round(stats_now["plan_s"], 4) if "plan_s" in stats_now else None),
                "streams": frame_loop.DEFAULT_STREAMS, "writer_threads": frame_loop.DEFAULT_WRITER_THREADS,
                "bytes_per_frame": int(nbytes), "png": frame_loop.png_mode(),
                "reference_shaped_loop": {"frames": len(ids), "ms_per_frame": round(t_ref / len(ids) * 1e3, 2),
                                          "frames_per_s": round(len(ids) / t_ref, 2), "bytes_per_frame": int(ref_bytes),
                                          "what": "per frame: deepcopy of the scene, the objects' PLYs from disk, transform + merge in PyTorch, one "
                                                  "blocking render() of this package, PIL / numpy host writers (one thread)"},
                "vs_reference_shaped_loop": round((frames / t_loop) / (len(ids) / t_ref), 1),
                **({"host_seconds": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in frame_loop.LAST_LOOP_STATS.items()}}
                   if frame_loop.LAST_LOOP_STATS else {}),
                **({"host_profile": host_profile} if host_profile else {})}
    finally:
        shutil.rmtree(root, ignore_errors=True)



# ---------------------------------------------------------------------------------------------------------------------
# also.c5_blend_frames: blender/blend_all.py::blend_frames as a whole -- file discovery, PNG / EXR decoding of every Blender layer
# at 2x the frame, resizes, smoke fill, composite, the frame's PNG -- through the drop-in and in the reference's shape
# ---------------------------------------------------------------------------------------------------------------------
def _synthetic_blender_tree(root, W, H, frames, distinct=4, seed=3):
    """<root>/scene/custom_camera_path/{images, traj/exp} + <root>/cache/out/{rgb,depth}_*: `distinct` different frames of layers at 2x
    (object, shadow catcher, object + shadow, 3DGS object, smoke, fire; EXR depth passes as Blender writes them: R = G = B = Z, half,
    ZIP), hard-linked under the names of `frames` frames."""
    from PIL import Image
    from autovfx_amd import exr
    g = np.random.default_rng(seed)
    results = os.path.join(root, "scene", "custom_camera_path", "traj", "exp")
    images = os.path.join(root, "scene", "custom_camera_path", "images")
    cache = os.path.join(root, "cache", "out")
    for d in (results, images):
        os.makedirs(d)
    cfg = os.path.join(root, "cfg.json")
    with open(cfg, "w") as f:
        json.dump({"blender_cache_dir": os.path.join(root, "cache"), "output_dir_name": "out"}, f)
    W2, H2 = 2 * W, 2 * H
    yy, xx = np.mgrid[0:H2, 0:W2].astype(np.float32)

    def blob(cx, cy, r):
        return np.clip(1.5 - np.hypot(xx - cx * W2, yy - cy * H2) / (r * H2), 0.0, 1.0)

    def rgba(alpha, tint):
        img = np.empty((H2, W2, 4), np.uint8)
        for c in range(3):
            img[..., c] = np.clip(tint[c] * (0.6 + 0.4 * np.sin(xx * 0.01 + c)) * 255, 0, 255).astype(np.uint8)
        img[..., 3] = (alpha * 255).astype(np.uint8)
        return img

    kinds_rgb = ("rgb_obj", "rgb_shadow", "rgb_all", "rgb_obj_3dgs", "rgb_smoke_fire", "rgb_smoke_fire_pre")
    kinds_d = ("depth_obj", "depth_shadow", "depth_obj_3dgs", "depth_smoke_fire")
    for k in range(distinct):
        sh = 0.03 * k
        a = {"rgb_obj": blob(0.45 + sh, 0.5, 0.2), "rgb_shadow": blob(0.5 + sh, 0.62, 0.3) * 0.6, "rgb_all": blob(0.47 + sh, 0.55, 0.3),
             "rgb_obj_3dgs": blob(0.7 - sh, 0.45, 0.12), "rgb_smoke_fire": blob(0.3 + sh, 0.35, 0.18) * 0.7, "rgb_smoke_fire_pre": blob(0.3 + sh, 0.4, 0.1)}
        bgimg = (g.integers(0, 255, (H, W, 4))).astype(np.uint8)
        bgimg[..., 3] = 255
        Image.fromarray(bgimg).save(os.path.join(images, f"src{k}.png"), compress_level=1)
        for j, kind in enumerate(kinds_rgb):
            os.makedirs(os.path.join(cache, kind), exist_ok=True)
            Image.fromarray(rgba(a[kind], (0.9 - 0.1 * j, 0.4 + 0.1 * j, 0.3 + 0.05 * j))).save(os.path.join(cache, kind, f"src{k}.png"), compress_level=1)
        for j, kind in enumerate(kinds_d):
            z = (3.0 + j + 2.0 * blob(0.5, 0.5, 0.5) + 0.2 * np.sin(yy * 0.02)).astype(np.float32)
            z[blob(0.5 + sh, 0.5, 0.45) <= 0] = 65504.0            # "nothing here"
            os.makedirs(os.path.join(cache, kind, f"src{k}"), exist_ok=True)
            # (zlib level 4: what OpenEXR 3.1.3+ -- Blender 3.x / 4.x -- compresses ZIP blocks with; earlier versions used 6)
            exr.write_exr(os.path.join(cache, kind, f"src{k}", "Image.exr"), {"R": z, "G": z, "B": z, "A": np.ones_like(z)}, half=True, level=4)
    for i in range(frames):
        k = i % distinct
        os.link(os.path.join(images, f"src{k}.png"), os.path.join(images, f"{i:05d}.png"))
        for kind in kinds_rgb:
            os.link(os.path.join(cache, kind, f"src{k}.png"), os.path.join(cache, kind, "{:0>3d}.png".format(i + 1)))
        for kind in kinds_d:
            d = os.path.join(cache, kind, "{:0>3d}".format(i + 1))
            os.makedirs(d)
            os.link(os.path.join(cache, kind, f"src{k}", "Image.exr"), os.path.join(d, "Image{:0>4d}.exr".format(i + 1)))
    for k in range(distinct):      # the sources themselves are not frames: rgb_all/*.png is what blend_frames counts
        os.remove(os.path.join(images, f"src{k}.png"))
        for kind in kinds_rgb:
            os.remove(os.path.join(cache, kind, f"src{k}.png"))
    return results, cfg


def _reference_shaped_blend_frames(results, cfg_path, frame_ids):
    """blend_all.blend_frames (:95-346) in the reference's shape, one host thread: PIL loads, the EXR reader (the reference's is
    cv2.imread, absent here), PIL resizes of every layer (:21-28,217-234), the numpy composite (oracle/compositor_oracle.py: pinned to
    the reference's function), Image.save of the frame."""
    from PIL import Image
    from autovfx_amd import compositor
    from oracle import compositor_oracle as co
    with open(cfg_path) as f:
        cfg = json.load(f)
    cache = os.path.join(cfg["blender_cache_dir"], cfg["output_dir_name"])
    root_dir = os.path.dirname(os.path.normpath(os.path.dirname(os.path.normpath(results))))
    out_dir = os.path.join(results, "frames_reference_shaped")
    os.makedirs(out_dir, exist_ok=True)
    names = {"o_c": "rgb_obj", "s_c": "rgb_shadow", "o_s_c": "rgb_all", "o_gs_c": "rgb_obj_3dgs", "s_f_c": "rgb_smoke_fire", "s_f_c_pre": "rgb_smoke_fire_pre",
             "o_d": "depth_obj", "s_d": "depth_shadow", "o_gs_d": "depth_obj_3dgs", "s_f_d": "depth_smoke_fire"}

    def down(a, size):
        if a.dtype == np.uint8:
            return np.array(Image.fromarray(a).resize(size, Image.BILINEAR))
        return np.array(Image.fromarray(a).resize(size, Image.NEAREST))

    for i in frame_ids:
        L = compositor._load_frame_layers(cache, os.path.join(root_dir, "images", f"{i:05d}.png"), i)
        args = {k: L[v] for k, v in names.items() if L[v] is not None}
        args["bg_c"] = L["bg"]
        if "s_f_c" in args:
            args["s_f_d"], _ = co.smoke_depth_fill(args["s_f_c"], args["s_f_d"].astype(np.float32), None)
        size = (args["bg_c"].shape[1], args["bg_c"].shape[0])
        for k in list(args):
            if k != "bg_c":
                args[k] = down(args[k] if args[k].dtype == np.uint8 else args[k].astype(np.float32), size)
        frame = co.composite_frame(**args)
        Image.fromarray(frame).save(os.path.join(out_dir, f"{i:04d}.png"))
    return out_dir


def blend_frames_bench(device, frames=200, reference_frames=3):
    """``autovfx_amd.compositor.blend_frames`` -- what ``install()`` puts behind ``blend_all.blend_frames`` -- on a synthetic
    Blender tree (960x540 frames, every Blender layer at 1920x1080: 6 RGBA PNGs and 4 half-float ZIP EXR depth passes per frame), to a
    tmpfs; the reference-shaped function beside it on a bounded number of frames."""
    import shutil
    import tempfile
    from PIL import Image
    from autovfx_amd import compositor
    W, H = 960, 540
    root = tempfile.mkdtemp(prefix="gsr_blend_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        results, cfg = _synthetic_blender_tree(root, W, H, frames)
        compositor.blend_frames(results, cfg, device=device, write_video=False)          # warm: resize tables, allocator, page cache
        shutil.rmtree(os.path.join(results, "frames"))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        paths = compositor.blend_frames(results, cfg, device=device, write_video=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ids = list(range(0, frames, max(1, frames // reference_frames)))[:reference_frames]
        t0 = time.perf_counter()
        ref_dir = _reference_shaped_blend_frames(results, cfg, ids)
        t_ref = time.perf_counter() - t0
        same = all(np.array_equal(np.asarray(Image.open(paths[i])), np.asarray(Image.open(os.path.join(ref_dir, f"{i:04d}.png")))) for i in ids)
        layer_bytes = sum(os.path.getsize(os.path.join(root, "cache", "out", k, "001.png")) for k in compositor._LAYERS_RGB) + \
            sum(os.path.getsize(os.path.join(root, "cache", "out", k, "001", "Image0001.exr")) for k in compositor._LAYERS_DEPTH)
        return {"workload": f"{frames} frames at 960x540; per frame 6 RGBA PNG layers + 4 half-float ZIP EXR depth passes at 1920x1080 (Blender at 2x), "
                            "the 3DGS frame's PNG; frames written as compressed PNGs to a tmpfs", "frames": len(paths),
                "value": round(len(paths) / dt, 1), "unit": "frames/s", "ms_per_frame": round(dt / len(paths) * 1e3, 3),
                "decode_threads": compositor.decode_threads(),
                **({"thread_seconds_by_stage": {k: round(v, 3) for k, v in compositor.LAST_BLEND_STATS.items() if k.endswith("_s")}}
                   if compositor.LAST_BLEND_STATS else {}),
                "input_bytes_per_frame": int(layer_bytes), "output_bytes_per_frame": int(os.path.getsize(paths[0])),
                "reference_shaped": {"frames": len(ids), "ms_per_frame": round(t_ref / len(ids) * 1e3, 1), "frames_per_s": round(len(ids) / t_ref, 3),
                                     "what": "one host thread: PIL loads, EXR reads, PIL resizes of ten layers, numpy composite, PIL save"},
                "same_pixels_as_reference_shaped": bool(same), "vs_reference_shaped": round((len(paths) / dt) / (len(ids) / t_ref), 1)}
    finally:
        shutil.rmtree(root, ignore_errors=True)



class ReferenceGetters:
    """A stand-in for the class AutoVFX instantiates (``sugar/gaussian_splatting/scene/gaussian_model.py:25-128``): the six
    raw parameter tensors, the activation functions kept as attributes (``setup_functions``) and getters that recompute on
    EVERY access -- no memo, nothing this repository added.  What ``render()`` sees when the caller is unchanged."""

    def __init__(self, cloud, sh_degree=3):
        from autovfx_amd.gaussian_model import inverse_sigmoid
        self.active_sh_degree = self.max_sh_degree = sh_degree
        self._xyz = cloud.means3D.clone()
        self._features_dc = cloud.shs[:, :1].clone().contiguous()
        self._features_rest = cloud.shs[:, 1:].clone().contiguous()
        self._scaling = torch.log(cloud.scales)
        self._rotation = (cloud.rotations * 1.7).contiguous()          # a trained model's quaternions are not unit length
        self._opacity = inverse_sigmoid(cloud.opacities.reshape(-1, 1).clamp(1e-6, 1 - 1e-6))
        self.scaling_activation, self.opacity_activation = torch.exp, torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    get_xyz = property(lambda self: self._xyz)
    get_scaling = property(lambda self: self.scaling_activation(self._scaling))
    get_rotation = property(lambda self: self.rotation_activation(self._rotation))
    get_opacity = property(lambda self: self.opacity_activation(self._opacity))
    get_features = property(lambda self: torch.cat((self._features_dc, self._features_rest), dim=1))

    @property
    def get_minimum_axis(self):
        from autovfx_amd.gaussian_model import get_minimum_axis
        return get_minimum_axis(self.get_scaling, self.get_rotation)

    def get_normal(self, dir_pp_normalized=None):
        from autovfx_amd.gaussian_model import flip_align_view
        normal_axis, _ = flip_align_view(self.get_minimum_axis, dir_pp_normalized)
        return normal_axis / normal_axis.norm(dim=1, keepdim=True)


def reference_shaped_render(device, key="c3", frames=24):
    """The per-frame ``render()`` as an UNCHANGED caller reaches it (scene_representation.py:424 after
    ``autovfx_amd.install()``): the model is the reference's class shape (``ReferenceGetters``: raw tensors + non-memoising
    getters), one blocking call per frame on one stream.

    ms_per_frame                 this repository's render() on that model: the raw-parameter path (gsr_forward_raw) -- no
                                 PyTorch activation, one rasterizer call, one elementwise kernel (round 4)
    params_changed_every_frame   the same with every parameter tensor edited in place before each frame (a training /
                                 dynamic-scene loop): nothing is cached between frames, so the rate must be the same
    pytorch_prep_ms_per_frame    the reference's own structure on this library: exp / sigmoid / normalize / cat / get_normal in
                                 PyTorch, two complete GaussianRasterizer calls, PyTorch post-processing (what round 3
                                 reported as ms_per_frame: 4.51); with the opt-in geometry reuse beside it
    fused_memo_ms_per_frame      rounds 2-3's fast path: this repository's memoising GaussianModel + gsr_view_normals
    The frames of the first two are bit-identical to the third's (tests/test_raw_gpu.py)."""
    from autovfx_amd import renderer
    from diff_gaussian_rasterization import _C
    b = Bench(key, device, None, boundary="render")
    ref_model = ReferenceGetters(b.cloud, b.cloud.sh_degree)
    fr = [(10 + 7 * j) % b.F for j in range(frames)]
    for f in fr:
        b.cam(f)
    state = {"model": ref_model, "edit": False}

    def run(n=frames):
        m = state["model"]
        with torch.no_grad():
            for f in fr[:n]:
                if state["edit"]:
                    m._xyz.add_(1e-6); m._scaling.add_(1e-6); m._rotation.mul_(1.0001); m._opacity.add_(1e-5)
                    m._features_dc.add_(1e-5); m._features_rest.mul_(0.9999)
                out = renderer.render(b.cam(f), m, renderer.PipelineParams, b.bg)
        return out

    def timed():
        run(4)
        secs = timed_regions(run, 3, False, device)
        return sorted(secs)[1] / frames * 1e3

    out = {"workload": b.name, "frames": frames, "streams": 1,
           "model": "reference-shaped GaussianModel stand-in: raw tensors, getters recompute on every access (bench.py: ReferenceGetters)"}
    out["raw_path_taken"] = renderer.raw_parameters(ref_model) is not None
    out["ms_per_frame"] = round(timed(), 4)
    state["edit"] = True
    out["params_changed_every_frame_ms_per_frame"] = round(timed(), 4)
    state["edit"] = False
    try:
        renderer.FUSE_ELEMENTWISE = False
        _C.set_geometry_cache(False)
        out["pytorch_prep_ms_per_frame"] = round(timed(), 4)
        _C.set_geometry_cache(True)
        out["pytorch_prep_with_geometry_reuse_opt_in_ms_per_frame"] = round(timed(), 4)
    finally:
        _C.set_geometry_cache(None)
        renderer.FUSE_ELEMENTWISE = True
    try:
        renderer.RAW_PARAMETERS = False
        state["model"] = b.model
        out["fused_memo_ms_per_frame"] = round(timed(), 4)
    finally:
        renderer.RAW_PARAMETERS = True
    out["value"] = round(1e3 / out["ms_per_frame"], 2)
    out["unit"] = "frames/s"
    out["speedup_vs_pytorch_prep"] = round(out["pytorch_prep_ms_per_frame"] / out["ms_per_frame"], 2)
    out["what"] = ("render() per frame for an unchanged caller: blocking, one stream, the reference's model class shape; "
                   "ms_per_frame = raw-parameter path (default), pytorch_prep = the reference's structure run through PyTorch "
                   "on this library (round 3's figure), bit-identical frames")
    return out


def training_render_iteration(device, key="c3", steps=10):
    """One iteration of the reference's training loops AS THEY CALL IT (scene_representation.py:495-520: ``render()`` with
    autograd on, an L1 loss on ``result["render"]``, ``loss.backward()``; no optimizer step) on the reference's model class
    shape (``ReferenceGetters``, leaves = the six raw tensors).  Two ways on this library:

    ms_per_iter                 render() differentiable from the raw tensors: one full rasterizer call (colour + normal image in
                                one walk), gsr_backward_raw with the activations' chain rule in the per-Gaussian kernel
    reference_structure_ms      renderer.RAW_AUTOGRAD = False: PyTorch activations + get_normal with their autograd graph, two
                                complete GaussianRasterizer calls, gsr_backward, autograd through the activations
    Same forward images bit for bit, gradients within the atomics tolerance (tests/test_raw_autograd_gpu.py)."""
    from autovfx_amd import renderer
    b = Bench(key, device, None, boundary="op")
    m = ReferenceGetters(b.cloud, b.cloud.sh_degree)
    params = ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc", "_features_rest")
    for k in params:
        setattr(m, k, getattr(m, k).detach().clone().requires_grad_(True))
    target = torch.rand(4, b.H, b.W, device=device)
    fr = [(10 + 7 * j) % b.F for j in range(steps)]
    for f in fr:
        b.cam(f)

    def run(n=steps):
        for f in fr[:n]:
            for k in params:
                getattr(m, k).grad = None
            out = renderer.render(b.cam(f), m, renderer.PipelineParams, b.bg)
            (out["render"] - target).abs().mean().backward()

    def timed():
        run(3)
        secs = timed_regions(run, 3, False, device)
        return sorted(secs)[1] / steps * 1e3

    out = {"workload": b.name, "steps": steps, "loss": "L1 on render() RGBA (scene_representation.py:507-510 without SSIM / LPIPS)"}
    out["ms_per_iter"] = round(timed(), 3)
    try:
        renderer.RAW_AUTOGRAD = False
        out["reference_structure_ms_per_iter"] = round(timed(), 3)
    finally:
        renderer.RAW_AUTOGRAD = True
    out["iters_per_s"] = round(1e3 / out["ms_per_iter"], 1)
    out["speedup_vs_reference_structure"] = round(out["reference_structure_ms_per_iter"] / out["ms_per_iter"], 2)
    return out


def backward_iteration(key, device, steps=12, parity=True):
    """One training-style iteration (train.py:84-134 without the optimizer): forward (a FULL call: gradients are wanted),
    L1 + depth loss, ``loss.backward()``; HIP events around the halves, the two backward kernels timed by the library's
    own events on the launch stream.  Gradient parity against the CPU oracle's backward on one frame."""
    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.frame_parallel import settings_for_camera
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    wl = WORKLOADS[key]
    W, H = wl["width"], wl["height"]
    cloud_cpu = getattr(scenes, wl["cfg"])()
    cloud = cloud_cpu.to(device)
    cams_cpu = orbit_cameras(wl["frames"], W, H)
    cams_dev = {i: cams_cpu[i].to(device) for i in range(0, 5 + steps + 1)}   # resident before any clock starts
    bg = torch.zeros(3, device=device)
    leaves = [t.clone().requires_grad_(True) for t in (cloud.means3D, cloud.opacities, cloud.shs, cloud.scales, cloud.rotations)]
    target = torch.rand(3, H, W, device=device)

    def it(i, timers=None):
        m3, op, sh, sc, rot = leaves
        for t in leaves:
            t.grad = None
        cam = cams_dev[i]
        rast = GaussianRasterizer(settings_for_camera(cam, bg, 3))
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        img, depth, alpha, radii = rast(m3, torch.zeros_like(m3, requires_grad=True), op, shs=sh, scales=sc, rotations=rot)
        loss = (img - target).abs().mean() + 0.01 * depth.mean()
        e[1].record()
        loss.backward()
        e[2].record()
        if timers is not None:
            timers.append(e)
        it.last_radii = radii
        return int((radii > 0).sum().item()) if timers is None else None

    for i in range(4):
        it(i)
    live = _C.last_layout()["counts"]["live_pairs"]
    torch.cuda.synchronize()
    timers = []
    t0 = time.perf_counter()
    for i in range(steps):
        it(5 + i, timers)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    # the two backward kernels by the library's own events, in a pass of their own: the per-stage events of a timed call (five per depth
    # slab) are not part of the iteration that is reported above
    _lib.set_stage_timing(True)
    for i in range(steps):
        it(5 + i, [])
    torch.cuda.synchronize()
    bw_k = _lib.backward_times_ms()
    _lib.set_stage_timing(False)
    fw = sum(a.elapsed_time(b_) for a, b_, _ in timers) / len(timers)
    bw = sum(b_.elapsed_time(c) for _, b_, c in timers) / len(timers)
    P = cloud.P
    # who the per-Gaussian kernel has work for (last iteration): rendered = reads its 64-byte line of sums; with a gradient = also
    # reads its parameters and coefficients (round 4: a rendered Gaussian whose sums are all zero only gets its zeros written)
    with torch.no_grad():
        m3, op, sh, sc, rot = leaves
        n_rendered = int((it.last_radii > 0).sum().item())
        n_active = int(((op.grad.reshape(P, -1) != 0).any(1) | (m3.grad != 0).any(1) | (sh.grad.reshape(P, -1) != 0).any(1)).sum().item())
    pb_moved = 252 * P + 64 * n_rendered + 232 * n_active   # 248 B of gradients + 4 B radius each; the line of sums; the inputs
    # render_backward_kernel: per list entry it walks the forward's 48-byte reads (id 4, raster 32, colour 12) and adds
    # one 40-byte line of partial sums per (entry, 8x8 quadrant) that contributes; per pixel 28 B in.  Algorithmic bytes
    # as SURVEY 8d counts the forward blend (44 B per pair + per-pixel terms) plus the ten 4-byte sums per pair the
    # reference's backward adds atomically: (44 + 40) * live pairs + 28 * W * H.
    rb_bytes = (44 + 40) * live + 28 * W * H
    out = {"workload": wl["name"], "P": P, "steps": steps, "iters_per_s": round(steps / el, 2), "ms_per_iter": round(el / steps * 1e3, 3),
           "forward_plus_loss_ms": round(fw, 3), "backward_ms": round(bw, 3), "live_pairs": int(live),
           "kernels": {"render_backward_kernel": {"ms": round(bw_k["render_backward"], 4), "alg_bytes": int(rb_bytes),
                                                  "alg_GBps": round(rb_bytes / max(1e-9, bw_k["render_backward"] * 1e-3) / 1e9, 1),
                                                  "frac_of_hbm_peak": round(rb_bytes / max(1e-9, bw_k["render_backward"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                                  "bound": "vector issue + LDS cross-lane reductions + one atomic line per (entry, quadrant)"},
                       # per Gaussian: inputs 12 + 12 + 16 + 192 (SH) + its 64-byte line of sums; gradients 12 (mean2D) + 4 (opacity) +
                       # 12 (mean3D) + 192 (SH) + 12 + 16 (scale, rotation) = 248 (round 4: conic 16, colour 12, depth 4 and cov3D 24 are
                       # not written when nobody reads them) -> 544 B
                       "preprocess_backward_kernel": {"ms": round(bw_k["preprocess_backward"], 4), "alg_bytes": int(544 * P),
                                                      "alg_GBps": round(544 * P / max(1e-9, bw_k["preprocess_backward"] * 1e-3) / 1e9, 1),
                                                      "frac_of_hbm_peak": round(544 * P / max(1e-9, bw_k["preprocess_backward"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                                      "alg_bytes_note": "544 B per Gaussian = what the reference's kernel moves for these gradients (an equivalence figure since "
                                                                        "round 4: Gaussians without a gradient are not read)",
                                                      "gaussians": {"rendered": n_rendered, "with_gradient": n_active},
                                                      "moved_bytes": int(pb_moved),
                                                      "moved_GBps": round(pb_moved / max(1e-9, bw_k["preprocess_backward"] * 1e-3) / 1e9, 1),
                                                      "frac_of_hbm_peak_moved": round(pb_moved / max(1e-9, bw_k["preprocess_backward"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                                      "bound": "hbm"}},
           "timed_calls": bw_k["calls"]}
    out["forward_call"] = "inference call (GSR_OPT_GRAD_SLABS: depth slabs, SH colours only for listed splats; the backward walks the slabs)" \
        if _lib.get_option(_lib.OPT_GRAD_SLABS) else "full call"
    out["forward_slab_pairs"] = _C.last_layout()["slab_pairs"]
    # the same iteration with the forward as a FULL call (GSR_OPT_GRAD_SLABS = 0: every live pair sorted, SH for every visible splat)
    try:
        _lib.set_option(_lib.OPT_GRAD_SLABS, 0)
        for i in range(3):
            it(i)
        torch.cuda.synchronize()
        timers = []
        t0 = time.perf_counter()
        for i in range(steps):
            it(5 + i, timers)
        torch.cuda.synchronize()
        el_f = time.perf_counter() - t0
        out["with_full_call_forward"] = {"ms_per_iter": round(el_f / steps * 1e3, 3),
                                         "forward_plus_loss_ms": round(sum(a.elapsed_time(b_) for a, b_, _ in timers) / len(timers), 3),
                                         "backward_ms": round(sum(b_.elapsed_time(c) for _, b_, c in timers) / len(timers), 3)}
    except Exception as e:
        out["with_full_call_forward"] = {"error": repr(e)[:200]}
    finally:
        _lib.set_option(_lib.OPT_GRAD_SLABS, 1)
    # GSR_OPT_BACKWARD_DETERMINISTIC: what a fixed summation order costs, and that it is one (the same frame twice: same bits)
    try:
        _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 1)
        for i in range(3):
            it(i)
        torch.cuda.synchronize()
        timers = []
        t0 = time.perf_counter()
        for i in range(steps):
            it(5 + i, timers)
        torch.cuda.synchronize()
        el_d = time.perf_counter() - t0
        it(5)
        first = [t.grad.clone() for t in leaves]
        it(5)
        same = all(bool(torch.equal(a, t.grad)) for a, t in zip(first, leaves))
        out["deterministic"] = {"ms_per_iter": round(el_d / steps * 1e3, 3),
                                "backward_ms": round(sum(b_.elapsed_time(c) for _, b_, c in timers) / len(timers), 3),
                                "same_bits_twice": same, "pool_bytes_per_pair": 160,
                                "what": "GSR_OPT_BACKWARD_DETERMINISTIC: per-Gaussian sums in a fixed order (records per (list position, "
                                        "quadrant), point list sorted by Gaussian, one lane per Gaussian) instead of float atomics"}
        del first
    except Exception as e:
        out["deterministic"] = {"error": repr(e)[:200]}
    finally:
        _lib.set_option(_lib.OPT_BACKWARD_DETERMINISTIC, 0)
    out["reference_on_gpu"] = reference_training_iteration(cloud, cams_cpu, bg, target, device, steps, cams_dev)
    if isinstance(out["reference_on_gpu"], dict) and "ms_per_iter" in out["reference_on_gpu"]:
        out["vs_reference_kernels"] = round(out["reference_on_gpu"]["ms_per_iter"] / out["ms_per_iter"], 2)
    if parity:
        try:
            from oracle import cpu_oracle
            g = np.random.default_rng(5)
            pg = dict(dL_dcolor=(g.standard_normal((3, H, W)) / (3 * H * W)).astype(np.float32),
                      dL_ddepth=np.full((1, H, W), 0.01 / (H * W), np.float32), dL_dalpha=np.zeros((1, H, W), np.float32))
            cam = cams_cpu[7]
            kw = dict(means3D=cloud_cpu.means3D, opacities=cloud_cpu.opacities, bg=np.zeros(3, np.float32), width=W, height=H,
                      viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,
                      tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, sh_degree=3, shs=cloud_cpu.shs, scales=cloud_cpu.scales,
                      rotations=cloud_cpu.rotations, **pg)
            t0 = time.perf_counter()
            ref, truth = cpu_oracle.backward(**kw), cpu_oracle.backward_f64(**kw)
            t_ref = time.perf_counter() - t0
            for t in leaves:
                t.grad = None
            camd = cam.to(device)
            img, depth, alpha, radii = GaussianRasterizer(settings_for_camera(camd, bg, 3))(
                leaves[0], torch.zeros_like(leaves[0], requires_grad=True), leaves[1], shs=leaves[2], scales=leaves[3], rotations=leaves[4])
            tt = lambda a: torch.from_numpy(a).to(device)
            ((img * tt(pg["dL_dcolor"])).sum() + (depth * tt(pg["dL_ddepth"])).sum() + (alpha * tt(pg["dL_dalpha"])).sum()).backward()
            torch.cuda.synchronize()
            rel, frac = {}, {}
            for name, leaf in zip(("dL_dmeans3D", "dL_dopacity", "dL_dsh", "dL_dscales", "dL_drotations"), leaves):
                a, r = leaf.grad.cpu().numpy().astype(np.float64).reshape(-1), ref[name].astype(np.float64).reshape(-1)
                t = truth[name].reshape(-1)
                scale = max(1e-30, float(np.abs(t).max()))
                e_hip, e_ref = float(np.abs(a - t).max()), float(np.abs(r - t).max())
                rel[name] = {"hip_vs_truth": float(f"{e_hip / scale:.3e}"), "reference_fp32_vs_truth": float(f"{e_ref / scale:.3e}"),
                             "hip_vs_reference_fp32": float(f"{np.abs(a - r).max() / scale:.3e}")}
                frac[name] = round(e_hip / max(2e-4 * scale + 1e-6, 4.0 * e_ref), 3)
            out["parity_vs_truth"] = {"frame": 7, "distance_over_scale": rel, "fraction_of_bar": frac,
                                      "bar": "max(2e-4 scale + 1e-6, 4 |reference_fp32 - truth|), truth = gsro_backward_f64 (fp64)",
                                      "radii_equal": bool((radii.cpu().numpy() == ref["radii"]).all()),
                                      "oracle_seconds": round(t_ref, 2)}
        except Exception as e:
            out["parity_vs_truth"] = {"error": repr(e)[:200]}
    return out


def reference_training_iteration(cloud, cams_cpu, bg, target, device, steps, cams_dev=None):
    """The same training-style iteration through the REFERENCE's own kernels compiled for gfx950 (oracle/ref_hip.py:
    forward.cu + backward.cu + rasterizer_impl.cu, hipCUB for its CUB calls): forward, the same loss gradients formed in
    PyTorch, backward.  A measuring stick: what "matching the reference" means for an iteration on this GPU.  The reference
    runs on the default stream with a host synchronisation inside each call; wall clock between synchronize()s."""
    try:
        from oracle import ref_hip
        if not ref_hip.available():
            return None
        H, W = int(target.shape[1]), int(target.shape[2])
        outs = (torch.zeros((3, H, W), device=device), torch.zeros((1, H, W), device=device), torch.zeros((1, H, W), device=device),
                torch.zeros((cloud.P,), dtype=torch.int32, device=device))
        dalpha = torch.zeros((1, H, W), device=device)

        def it(i, t):
            cam = cams_dev[i] if cams_dev is not None and i in cams_dev else cams_cpu[i].to(device)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n, color, depth, alpha, radii = ref_hip.forward(cloud, cam, bg, outs)
            t1 = time.perf_counter()
            # d/d(img) of (img - target).abs().mean() + 0.01 * depth.mean(), as autograd would hand it to the backward
            dcolor = torch.sign(color - target) / color.numel()
            ddepth = torch.full_like(depth, 0.01 / depth.numel())
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ref_hip.backward(cloud, cam, bg, n, radii, alpha, dcolor, ddepth, dalpha)
            t3 = time.perf_counter()
            if t is not None:
                t.append((t1 - t0, t2 - t1, t3 - t2))

        for i in range(3):
            it(i, None)
        t = []
        for i in range(steps):
            it(5 + i, t)
        fw, loss, bw = (sum(x[k] for x in t) / len(t) * 1e3 for k in range(3))
        return {"ms_per_iter": round(fw + loss + bw, 3), "forward_ms": round(fw, 3), "loss_grad_ms": round(loss, 3),
                "backward_ms": round(bw, 3), "steps": steps,
                "kind": "reference sources (forward.cu, backward.cu, rasterizer_impl.cu; hipCUB for its CUB calls) compiled for "
                        "gfx950, -ffp-contract=off; gradient buffers allocated and zero-filled per call as its binding does"}
    except Exception as e:   # a measuring stick must not break the line
        return {"error": repr(e)[:200]}


def time_frame_files(b, n):
    """What the reference writes per frame (scene_representation.py:425-438: RGBA PNG, depth .npy, normal PNG), timed
    separately from the render + composite rate: host-side zlib + file system, on `n` frames."""
    import tempfile
    from autovfx_amd import renderer
    from autovfx_amd.frame_io import write_frame_outputs
    from autovfx_amd.gaussian_model import GaussianModel
    try:
        c = b.cloud
        model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, c.sh_degree)
        with torch.no_grad(), tempfile.TemporaryDirectory() as d:
            t_total = 0.0
            for j in range(n):
                res = renderer.render(b.cam(j * 7), model, renderer.PipelineParams, b.bg)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                write_frame_outputs(d, f"{j:05d}", res)
                t_total += time.perf_counter() - t0
            # the same files through the pool of host threads scripts/render_trajectory.py uses (FrameWriter)
            from autovfx_amd.frame_io import FrameWriter
            m = 4 * n
            frames = [renderer.render(b.cam(j * 7), model, renderer.PipelineParams, b.bg) for j in range(4)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with FrameWriter(d) as w:
                workers = w._pool._max_workers
                for j in range(m):
                    w.submit(f"p{j:05d}", frames[j % 4])
            t_pool = time.perf_counter() - t0
            # ... and with the file images built on the GPU (gsr_png_encode: stored deflate, Adler-32 / CRC-32 in the kernel; the
            # .npy header in front of the depth plane), one D2H copy per frame, four host threads that only write()
            from autovfx_amd.frame_io import GpuFrameWriter
            mg = 16 * n
            gpu = {}
            for mode, deflate in (("compressed", True), ("stored", False)):
                with GpuFrameWriter(d, workers=4, deflate=deflate) as w:
                    for j in range(8):
                        w.submit(f"w{j:05d}", frames[j % 4])     # slots, pinned buffers, the first writes
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                with GpuFrameWriter(d, workers=4, deflate=deflate) as w:
                    for j in range(mg):
                        w.submit(f"g{j:05d}", frames[j % 4])
                t_gpu = time.perf_counter() - t0
                file_bytes = sum(os.path.getsize(os.path.join(d, sub, "g00000" + ext)) for sub, ext in
                                 (("images", ".png"), ("depth", ".npy"), ("depth", ".png"), ("normal", ".png")))
                png_bytes = file_bytes - os.path.getsize(os.path.join(d, "depth", "g00000.npy"))
                gpu[mode] = {"frames": mg, "threads": 4, "ms_per_frame": round(t_gpu / mg * 1e3, 3), "frames_per_s": round(mg / t_gpu, 1),
                             "bytes_per_frame": int(file_bytes), "png_bytes_per_frame": int(png_bytes),
                             "GBps_to_files": round(file_bytes * mg / t_gpu / 1e9, 2)}
            # what PIL (torchvision.utils.save_image's writer) makes of the same three images at its default level: the size yardstick
            import io as _io
            from PIL import Image as _Image
            from autovfx_amd.frame_io import _frame_to_host, depth2img
            rgba8, depth_h, normal_h = _frame_to_host(frames[0])
            pil_bytes = 0
            for img in (rgba8, depth2img(depth_h.squeeze(), 3.0), normal_h):
                buf = _io.BytesIO()
                _Image.fromarray(np.ascontiguousarray(img)).save(buf, format="PNG")
                pil_bytes += buf.getbuffer().nbytes
            gpu["compressed"]["png_bytes_vs_pil_default"] = round(gpu["compressed"]["png_bytes_per_frame"] / pil_bytes, 3)
            gpu["pil_default_png_bytes_per_frame"] = int(pil_bytes)
            gpu["what"] = ("the same four files, their bytes built on the GPU -- compressed: Paeth-filtered scanlines, run-length matches, one Huffman "
                           "code per image built in the kernel (gsr_frame_files_deflate); stored: deflate stored blocks (gsr_frame_files); "
                           "Adler-32 / CRC-32 in the kernels, .npy header + fp32 plane -- one D2H copy per frame into pinned memory, four host "
                           "threads write() to a temporary directory")
        return {"frames": n, "ms_per_frame": round(t_total / n * 1e3, 2),
                "what": "RGBA PNG + depth .npy + depth preview PNG + normal PNG per frame (D2H copies, zlib level 3, file writes), one host thread",
                "writer_pool": {"frames": m, "threads": workers, "ms_per_frame": round(t_pool / m * 1e3, 2)},
                "gpu_file_images": gpu}
    except Exception as e:
        return {"error": repr(e)[:200]}


def stage_counter_bytes(traffic):
    """HBM bytes per frame and stage from the committed counter summary: kernels are attributed to stages by name; the
    three radix kernels serve both sorts, so their bytes are split by the summary's own per-sort entries when it has them
    (pmc_reduce.py writes `stages`), else left out."""
    if traffic is None:
        return None
    if "stages" in traffic:
        return {k: float(v) for k, v in traffic["stages"].items()}
    out = {}
    for stage, names in STAGE_KERNELS.items():
        tot = 0.0
        for n in names:
            k = traffic["kernels"].get(n)
            if k is not None:
                tot += k.get("bytes_per_frame", k["bytes_per_launch"] * k.get("launches_per_frame", 1))
        if names:
            out[stage] = tot
    return out


def pmc_traffic(T_key):
    """HBM-side bytes per launch of each kernel and per frame, from the newest committed rocprofv3 PMC summary
    (profiles/r*_traffic.json, written by scripts/pmc_reduce.py from separate --pmc passes of `bench.py --profile-run
    --streams 1` on the C3 workload): 2 * FETCH_SIZE + WRITE_SIZE in KiB -- every L2 miss is a 128-byte request that
    FETCH_SIZE tallies at 64 bytes (calibrated on known byte counts, streaming AND random-gather patterns:
    profiles/r02_pmc_calibration.csv).  Stamped with the commit the passes were taken at.  None if there is none."""
    if T_key != "c3":
        return None
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not paths:
        return None
    try:
        with open(paths[-1]) as f:
            t = json.load(f)
        t["source"] = os.path.relpath(paths[-1], ROOT) + " (2*FETCH_SIZE + WRITE_SIZE, KiB; separate --pmc passes)"
        return t
    except (OSError, ValueError, KeyError):
        return None


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


def torch_cpu_frame(cfg, W, H, F, frame, threads, timeout_s, label):
    """The PyTorch-CPU restatement of the splat (oracle/torch_splat.py: the "CPU-only PyTorch splat path" of the north
    star) on ONE frame of a workload, in a subprocess with a hard time limit and a bounded thread count (with one thread
    per core of a 256-thread host its many small tensor ops take minutes, not seconds).  Seconds per frame as measured;
    a frame that does not finish inside the limit is reported as such, not extrapolated."""
    code = (
        "import sys, time, json, torch; sys.path.insert(0, %r); torch.set_num_threads(%d)\n"
        "from oracle import torch_splat; from autovfx_amd import scenes; from autovfx_amd.cameras import orbit_cameras\n"
        "cloud = scenes.%s()\n"
        "cam = scenes.c1_camera() if %r == 'config_c1' else orbit_cameras(%d, %d, %d)[%d]\n"
        "kw = dict(means3D=cloud.means3D, opacities=cloud.opacities, width=cam.image_width, height=cam.image_height,"
        " viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform, campos=cam.camera_center,"
        " tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, sh_degree=cloud.sh_degree, scale_modifier=1.0, shs=cloud.shs,"
        " scales=cloud.scales, rotations=cloud.rotations)\n"
        "t0 = time.perf_counter(); torch_splat.forward(bg=torch.zeros(3), **kw); dt = time.perf_counter() - t0\n"
        "print(json.dumps({'s': dt}))\n") % (ROOT, threads, cfg, cfg, F, W, H, frame)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s)
        s = json.loads(r.stdout.strip().splitlines()[-1])["s"]
        return {"seconds_per_frame": round(s, 3), "value": round(1.0 / s, 4), "unit": "frames/s", "cores": threads,
                "sample": f"{label}, frame {frame}, oracle/torch_splat.py, one run"}
    except subprocess.TimeoutExpired:
        return {"timeout": True, "limit_s": timeout_s, "cores": threads, "sample": f"{label}, frame {frame}: not finished after "
                f"{time.perf_counter() - t0:.0f} s (no extrapolation)"}
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


def run_cpu_baseline(b, budget_s=12.0, max_frames=12):
    """Time the CPU oracle (OpenMP over the CPUs this container may use) on a bounded sample of the same workload --
    the first, middle and last frame of the orbit, then more frames until ~budget_s of wall time -- and report the
    GPU's parity on those three; then the PyTorch-CPU splat on one frame of C1, C2 and of this workload."""
    from oracle import cpu_oracle
    from autovfx_amd.frame_parallel import rasterize
    logical, usable = host_cpu_budget()
    threads = max(1, int(round(usable)))
    try:   # one OpenMP thread per usable CPU: 256 threads inside a 16-CPU quota only fight each other
        import ctypes
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    except OSError:
        threads = logical
    c, W, H, F = b.cloud_cpu, b.W, b.H, b.F

    def kw(cam):
        return dict(means3D=c.means3D, opacities=c.opacities, bg=np.zeros(3, np.float32), width=W,
                    height=H, viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
                    campos=cam.camera_center, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
                    sh_degree=c.sh_degree, shs=c.shs, scales=c.scales, rotations=c.rotations)

    cpu_oracle.lib()
    parity_frames = [0, F // 2, F - 1]
    sample = parity_frames + [(20 + 7 * n) % F for n in range(max_frames)]
    total, n, parity = 0.0, 0, []
    for f in sample:
        if n >= len(parity_frames) and (n >= max_frames or total >= budget_s):
            break
        t0 = time.perf_counter()
        ref = cpu_oracle.forward(**kw(b.cams_cpu[f]))
        total += time.perf_counter() - t0
        n += 1
        if f in parity_frames and len(parity) < len(parity_frames):
            with torch.no_grad():
                color, depth, alpha, radii = rasterize(b.cloud, b.cam(f), b.bg)
                # the same frame through the FULL (differentiable) call -- no depth slabs, no deferred colours, the lists the
                # backward reads: the inference call the timed region makes must give the same bits
                from diff_gaussian_rasterization import _C
                cam_f, e_ = b.cam(f), torch.Tensor([])
                full = _C.rasterize_gaussians(b.bg, b.cloud.means3D, e_, b.cloud.opacities, b.cloud.scales, b.cloud.rotations, 1.0, e_,
                                              cam_f.world_view_transform, cam_f.full_proj_transform, cam_f.tanfovx, cam_f.tanfovy,
                                              int(cam_f.image_height), int(cam_f.image_width), b.cloud.shs, b.cloud.sh_degree,
                                              cam_f.camera_center, False, False, inference=False)
            torch.cuda.synchronize()
            same = all(bool(torch.equal(x, y)) for x, y in zip((color, depth, alpha, radii), full[1:5]))
            err = np.abs(color.cpu().numpy() - ref["color"]).max(axis=0)
            parity.append({"frame": f, "rgb_maxabs": float(err.max()), "rgb_px_over_1e-4": int((err > 1e-4).sum()),
                           "alpha_maxabs": float(np.abs(alpha.cpu().numpy() - ref["alpha"]).max()),
                           "depth_maxabs": float(np.abs(depth.cpu().numpy() - ref["depth"]).max()),
                           "radii_equal": bool((radii.cpu().numpy() == ref["radii"]).all()), "pixels": int(W * H),
                           "full_call_same_bits": same})
            del full
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    wl = WORKLOADS[b.key]
    out = {"value": round(n / total, 4), "unit": "frames/s", "cores": threads, "kind": "port",
           "seconds_per_frame": round(total / n, 3),
           "sample": f"{n} frames of the same workload at full size (orbit frames {sample[:n]}), C+OpenMP oracle, "
                     f"{total:.2f} s of wall time on {threads} OpenMP threads ({logical} logical CPUs, "
                     f"cgroup quota {usable:g} CPUs)",
           "cpu_model": model, "parity": parity,
           "torch_cpu_c1": torch_cpu_frame("config_c1", 256, 256, 1, 0, threads, 30.0, "C1 (10k Gaussians, 256x256)"),
           "torch_cpu_c2": torch_cpu_frame("config_c2", 960, 540, 200, 100, threads, 60.0, "C2 (1M Gaussians, 960x540)")}
    if b.key == "c3":
        out["torch_cpu_c3"] = torch_cpu_frame(wl["cfg"], W, H, F, F // 2, threads, 150.0, "C3 (3M Gaussians, 1920x1080)")
        # north_star's "reference's CPU-only PyTorch splat path timed on the host cores of the same box", as a scalar at the top
        # level of cpu_baseline (a flattening reader keeps it): seconds per C3 frame, null if the frame did not finish in 150 s
        out["torch_cpu_c3_seconds_per_frame"] = out["torch_cpu_c3"].get("seconds_per_frame")
    out["parity_full_call_same_bits"] = all(p.get("full_call_same_bits", False) for p in parity)
    out["parity_rgb_maxabs"] = max((p["rgb_maxabs"] for p in parity), default=None)
    return out


if __name__ == "__main__":
    main()
