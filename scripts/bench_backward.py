#!/usr/bin/env python
"""Times forward + backward (one training-style iteration without the optimizer) on a bench workload.
Not the headline metric (that is bench.py); used to profile the backward kernels."""
import argparse, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
from autovfx_amd.frame_parallel import settings_for_camera
from diff_gaussian_rasterization import GaussianRasterizer

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c2")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--reference", action="store_true",
                help="time the REFERENCE's own forward+backward kernels compiled for gfx950 (oracle/_ref/libgsr_ref_hip.so)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = {"c2": (scenes.config_c2, 960, 540, 200), "c3": (scenes.config_c3, 1920, 1080, 800),
       "heavy": (scenes.config_heavy, 960, 540, 200), "heavy1080": (scenes.config_heavy, 1920, 1080, 200)}[args.workload]
cloud = cfg[0]().to(dev)
cams = [c.to(dev) for c in orbit_cameras(cfg[3], cfg[1], cfg[2])[:args.steps + 5]]
bg = torch.zeros(3, device=dev)
if args.reference:
    from oracle import ref_hip
    g = torch.Generator(device=dev).manual_seed(0)
    H, W = cfg[2], cfg[1]
    wc, wd, wa = (torch.randn((3, H, W), generator=g, device=dev) / (3 * H * W), torch.full((1, H, W), 0.01 / (H * W), device=dev),
                  torch.zeros((1, H, W), device=dev))
    def ref_it(i):
        n, c, d, a, r = ref_hip.forward(cloud, cams[i], bg)
        t1 = time.perf_counter()
        ref_hip.backward(cloud, cams[i], bg, n, r, a, wc, wd, wa)
        return time.perf_counter() - t1
    for i in range(3):
        ref_it(i)
    t0 = time.perf_counter()
    bw = sum(ref_it(5 + i) for i in range(args.steps))
    el = time.perf_counter() - t0
    print(json.dumps({"workload": args.workload, "P": cloud.P, "kind": "reference kernels on this GPU (incl. zero-filling its gradient buffers)",
                      "ms_per_iter": round(el / args.steps * 1e3, 3), "backward_ms": round(bw / args.steps * 1e3, 3)}))
    sys.exit(0)

leaves = [t.clone().requires_grad_(True) for t in (cloud.means3D, cloud.opacities, cloud.shs, cloud.scales, cloud.rotations)]
target = torch.rand(3, cfg[2], cfg[1], device=dev)

def it(i, timers=None):
    m3, op, sh, sc, rot = leaves
    for t in leaves:
        t.grad = None
    rast = GaussianRasterizer(settings_for_camera(cams[i], bg, 3))
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    img, depth, alpha, radii = rast(m3, torch.zeros_like(m3, requires_grad=True), op, shs=sh, scales=sc, rotations=rot)
    loss = (img - target).abs().mean() + 0.01 * depth.mean()
    e[1].record()
    loss.backward()
    e[2].record()
    if timers is not None:
        timers.append(e)

for i in range(5):
    it(i)
torch.cuda.synchronize()
timers = []
t0 = time.perf_counter()
for i in range(args.steps):
    it(5 + i, timers)
torch.cuda.synchronize()
el = time.perf_counter() - t0
fw = sum(a.elapsed_time(b) for a, b, _ in timers) / len(timers)
bw = sum(b.elapsed_time(c) for _, b, c in timers) / len(timers)
print(json.dumps({"workload": args.workload, "P": cloud.P, "iters_per_s": round(args.steps / el, 2),
                  "ms_per_iter": round(el / args.steps * 1e3, 3), "forward_plus_loss_ms": round(fw, 3),
                  "backward_ms": round(bw, 3)}))
