"""Census of the blend kernel's inner loop (trace build): how many list entries a quadrant wave sees, how many
pass the quadrant reach test, how many have a live pixel, how many lanes are live.
    python -m autovfx_amd.build --trace
    GSR_LIB=autovfx_amd/lib/libgsr_hip_trace.so python scripts/blend_census.py [--workload c3]
"""
import argparse, ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="c3"); args = ap.parse_args()
    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.frame_parallel import rasterize
    import bench
    wl = bench.WORKLOADS[args.workload]
    cloud = getattr(scenes, wl["cfg"])().to("cuda")
    cam = orbit_cameras(wl["frames"], wl["width"], wl["height"])[0].to("cuda")
    bg = torch.zeros(3, device="cuda")
    rasterize(cloud, cam, bg); torch.cuda.synchronize()
    trace = torch.zeros(64, dtype=torch.int64, device="cuda")
    _lib.lib.gsr_debug_set_trace.argtypes = [ctypes.c_void_p]
    _lib.lib.gsr_debug_set_trace(trace.data_ptr())
    rasterize(cloud, cam, bg); torch.cuda.synchronize()
    _lib.lib.gsr_debug_set_trace(None)
    listed, reach, anylive, live_lanes, cand_lanes, waves = [int(v) for v in trace[:6].cpu()]
    print(f"quadrant waves {waves}; list entries seen {listed} ({listed / waves:.1f}/wave)")
    print(f"pass quadrant reach test {reach} ({100 * reach / listed:.1f}% of seen)")
    print(f"have a live pixel {anylive} ({100 * anylive / max(reach, 1):.1f}% of reaching)")
    it4, it2, sub_total = [int(v) for v in trace[6:9].cpu()]
    print(f"iterations if 4 sub-blocks of 4x4 walk their own lists: {it4} ({100 * it4 / max(reach, 1):.1f}% of now); "
          f"2 halves of 8x4: {it2} ({100 * it2 / max(reach, 1):.1f}%); mean sub-blocks reached per processed entry {sub_total / max(reach, 1):.2f}")
    print(f"live lanes {live_lanes}: {live_lanes / max(anylive, 1):.1f} of 64 per live entry; not-done lanes per processed entry {cand_lanes / max(reach, 1):.1f}")

if __name__ == "__main__":
    main()
