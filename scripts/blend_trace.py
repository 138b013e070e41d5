"""Per-wave timeline of the blend kernel (trace build): how long each quadrant wave lives, how that relates to
its list length, and how many waves are resident over the launch.
    python -m autovfx_amd.build --trace
    GSR_LIB=autovfx_amd/lib/libgsr_hip_trace.so python scripts/blend_trace.py [--variant 1]
"""
import argparse, ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="c3"); ap.add_argument("--variant", type=int, default=1)
    args = ap.parse_args()
    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.frame_parallel import rasterize
    import bench
    _lib.set_option(_lib.OPT_BLEND_VARIANT, args.variant)
    wl = bench.WORKLOADS[args.workload]
    cloud = getattr(scenes, wl["cfg"])().to("cuda")
    cam = orbit_cameras(wl["frames"], wl["width"], wl["height"])[0].to("cuda")
    bg = torch.zeros(3, device="cuda")
    rasterize(cloud, cam, bg); torch.cuda.synchronize()
    trace = torch.zeros((1 << 16) * 8, dtype=torch.int64, device="cuda")
    _lib.lib.gsr_debug_set_trace.argtypes = [ctypes.c_void_p]
    _lib.lib.gsr_debug_set_trace(trace.data_ptr())
    rasterize(cloud, cam, bg); torch.cuda.synchronize()
    _lib.lib.gsr_debug_set_trace(None)
    t = trace.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 4] > 0]
    start, end, count = t[:, 4] * 0.01, t[:, 5] * 0.01, t[:, 6]   # the blend stamps slots 4, 5, 6 of its record
    t0 = start.min(); start -= t0; end -= t0
    dur = end - start
    print(f"variant {args.variant}: waves {len(t)}  span {end.max():.1f} us  sum of wave lives {dur.sum() / 1e3:.1f} ms  "
          f"(= {dur.sum() / end.max():.0f} waves resident on average, of 8192 slots)")
    print("wave life us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
    print("list length: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d" % (count.mean(), *np.percentile(count, [50, 90, 99]), count.max()))
    heavy = np.argsort(-dur)[:5]
    for i in heavy:
        print(f"   wave life {dur[i]:.1f} us  list {count[i]}  start {start[i]:.1f}  ({dur[i] * 1e3 / max(count[i], 1):.0f} ns per listed entry)")
    edges = np.linspace(0, end.max(), 11)
    res = [int(((start < b) & (end > a)).sum()) for a, b in zip(edges[:-1], edges[1:])]
    print("waves resident per tenth of the launch:", res)
    print("starts per tenth:", [int(((start >= a) & (start < b)).sum()) for a, b in zip(edges[:-1], edges[1:])])

if __name__ == "__main__":
    main()
