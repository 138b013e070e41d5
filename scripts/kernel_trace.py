"""Per-workgroup phase timing of an instrumented kernel (libgsr_hip_trace.so, GSR_KERNEL_TRACE).

    python -m autovfx_amd.build --trace
    GSR_LIB=autovfx_amd/lib/libgsr_hip_trace.so python scripts/kernel_trace.py [--workload c3] [--slots 4]

Renders a few frames of the bench workload, arms the trace buffer, renders one more and prints, for the
workgroups that stamped slot 0: start-time quantiles, the kernel's span and the mean time between stamps.
Which kernel is instrumented is decided in the source (GSR_KTRACE calls); one kernel at a time.
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--slots", type=int, default=4)
    ap.add_argument("--records", type=int, default=1 << 17)
    ap.add_argument("--binning", action="store_true",
                    help="the binning kernels' stamps (gsr_binning.hip GSR_BTRACE): bin_gather, slab_recount, expand of slab 0 / 1")
    ap.add_argument("--colour", action="store_true", help="sh_colour_listed_kernel's stamps (gsr_kernels.hip)")
    ap.add_argument("--blend", action="store_true",
                    help="blend_quadrant_kernel: start / end of every single-wave workgroup -> waves in flight over time, the tail")
    args = ap.parse_args()
    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.frame_parallel import rasterize
    import bench
    wl = bench.WORKLOADS[args.workload]
    cloud = getattr(scenes, wl["cfg"])().to("cuda")
    cam = orbit_cameras(wl["frames"], wl["width"], wl["height"])[0].to("cuda")
    bg = torch.zeros(3, device="cuda")
    for _ in range(3):
        rasterize(cloud, cam, bg)
    torch.cuda.synchronize()
    trace = torch.zeros(args.records * 8, dtype=torch.int64, device="cuda")
    arm = (_lib.lib.gsr_debug_set_binning_trace if args.binning else _lib.lib.gsr_debug_set_blend_trace if args.blend
           else _lib.lib.gsr_debug_set_trace)
    arm.argtypes = [ctypes.c_void_p]
    assert arm(trace.data_ptr()) == 0
    rasterize(cloud, cam, bg)
    torch.cuda.synchronize()
    arm(None)
    t = trace.cpu().numpy().reshape(-1, 8).astype(np.float64)
    def groups(specs):
        for name, base, slots in specs:
            g = t[base:base + 8192]
            started = g[g[:, 0] > 0] * 0.01
            g = started[started[:, slots - 1] > 0][:, :slots]   # workgroups that ran to the end (the others left at once)
            if len(g) == 0:
                print(name, "no records"); continue
            t0 = started[:, 0].min()
            print(f"{name}: {len(started)} workgroups started, {len(g)} ran through; span {g.max() - t0:.1f} us; "
                  f"starts p50 {np.median(g[:, 0]) - t0:.1f} p95 {np.percentile(g[:, 0], 95) - t0:.1f} max {g[:, 0].max() - t0:.1f} us")
            for k in range(slots - 1):
                d = g[:, k + 1] - g[:, k]
                print(f"   phase {k}->{k + 1}: mean {d.mean():6.2f} us  p50 {np.median(d):6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f}")
            life = g[:, slots - 1] - g[:, 0]
            print(f"   workgroup life: mean {life.mean():.2f} us  p95 {np.percentile(life, 95):.2f}  max {life.max():.2f}")

    if args.blend:
        rec = trace.cpu().numpy().reshape(-1, 2).astype(np.float64) * 0.01
        for slab in (0, 1):
            g = rec[slab * 40960:(slab + 1) * 40960]
            g = g[g[:, 0] > 0]
            if len(g) == 0:
                continue
            ran = g[g[:, 1] > 0]
            if len(ran) == 0:
                continue
            t0, t1 = g[:, 0].min(), ran[:, 1].max()
            life = ran[:, 1] - ran[:, 0]
            print(f"blend launch {slab}: {len(g)} waves started, {len(ran)} stamped their end; span {t1 - t0:.1f} us; wave life mean {life.mean():.1f} "
                  f"p50 {np.median(life):.1f} p95 {np.percentile(life, 95):.1f} max {life.max():.1f} us; sum of lives / (8192 slots x span) = "
                  f"{life.sum() / (8192 * (t1 - t0)):.3f}")
            # waves in flight over time, in 20 steps of the span
            edges = np.linspace(t0, t1, 21)
            mid = 0.5 * (edges[1:] + edges[:-1])
            inflight = [(int(((ran[:, 0] <= m) & (ran[:, 1] > m)).sum())) for m in mid]
            print("   waves in flight at the middle of each 5 % of the span:", inflight)
            full = rec[slab * 40960:(slab + 1) * 40960]
            idx = np.flatnonzero((full[:, 0] > 0) & (full[:, 1] > 0))
            for x in range(8):   # workgroup b runs on XCD b % 8
                m = idx[idx % 8 == x]
                lv = full[m, 1] - full[m, 0]
                print(f"   XCD {x}: {len(m)} waves, sum of lives {lv.sum() / 1e3:.2f} ms, last end at {full[m, 1].max() - t0:.1f} us, "
                      f"last start at {full[m, 0].max() - t0:.1f} us")
            print("   last start at %.1f us; time with fewer than 4096 waves in flight at the end: %.1f us" % (
                ran[:, 0].max() - t0, t1 - next((m for m, n in zip(mid[::-1], inflight[::-1]) if n >= 4096), t0)))
        return
    if args.binning:
        groups((("bin_gather_kernel", 0, 5), ("slab_recount_kernel", 8192, 6), ("expand_kernel slab 0", 16384, 5),
                ("expand_kernel slab 1", 24576, 5)))
        return
    if args.colour:   # (lane 0 of the workgroup = its first wave: the other three waves are not stamped)
        groups((("sh_colour_listed_kernel slab 0", 16384, 4), ("sh_colour_listed_kernel slab 1", 24576, 4)))
        return
    t[16384:] = 0   # (the projection kernel's records)
    t = t[t[:, 0] > 0][:, :args.slots] * 0.01          # 100 MHz -> microseconds
    t0 = t[:, 0].min()
    q = [0, 25, 50, 75, 100]
    print(f"workgroups {len(t)}  span {t.max() - t0:.1f} us")
    print("start quantiles (by record index) us:", " ".join(f"{t[(len(t) - 1) * p // 100, 0] - t0:.1f}" for p in q))
    for k in range(args.slots - 1):
        d = t[:, k + 1] - t[:, k]
        print(f"phase {k}->{k + 1}: mean {d.mean():.2f} us  p50 {np.median(d):.2f}  p95 {np.percentile(d, 95):.2f}  max {d.max():.2f}")
    life = t[:, args.slots - 1] - t[:, 0]
    print(f"workgroup life: mean {life.mean():.2f} us  max {life.max():.2f}")


if __name__ == "__main__":
    main()
