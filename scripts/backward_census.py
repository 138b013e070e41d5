"""Census of render_backward_kernel's inner loop (trace build): how many (quadrant, entry) iterations reach each stage of
the loop and how many of the wave's 64 lanes contribute to an entry -- the lane efficiency of the gradient block and of the
cross-lane reduction.
    python -m autovfx_amd.build --trace
    GSR_LIB=autovfx_amd/lib/libgsr_hip_trace.so python scripts/backward_census.py [--workload c3] [--frame 7]
"""
import argparse, ctypes, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frame", type=int, default=7)
    args = ap.parse_args()
    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from autovfx_amd.frame_parallel import settings_for_camera
    from diff_gaussian_rasterization import GaussianRasterizer, _C
    import bench
    wl = bench.WORKLOADS[args.workload]
    dev = "cuda:0"
    cloud = getattr(scenes, wl["cfg"])().to(dev)
    cam = orbit_cameras(wl["frames"], wl["width"], wl["height"])[args.frame].to(dev)
    bg = torch.zeros(3, device=dev)
    leaves = [t.clone().requires_grad_(True) for t in (cloud.means3D, cloud.opacities, cloud.shs, cloud.scales, cloud.rotations)]
    target = torch.rand(3, wl["height"], wl["width"], device=dev)

    def it():
        for t in leaves:
            t.grad = None
        m3, op, sh, sc, rot = leaves
        img, depth, alpha, radii = GaussianRasterizer(settings_for_camera(cam, bg, 3))(m3, torch.zeros_like(m3, requires_grad=True), op,
                                                                                        shs=sh, scales=sc, rotations=rot)
        ((img - target).abs().mean() + 0.01 * depth.mean()).backward()
        torch.cuda.synchronize()

    it()
    live = _C.last_layout()["counts"]["live_pairs"]
    words = torch.zeros(16, dtype=torch.int64, device=dev)
    _lib.lib.gsr_debug_set_backward_census.argtypes = [ctypes.c_void_p]
    _lib.lib.gsr_debug_set_backward_census(words.data_ptr())
    it()
    _lib.lib.gsr_debug_set_backward_census(None)
    w = [int(v) for v in words.cpu()]
    T = ((wl["width"] + 15) // 16) * ((wl["height"] + 15) // 16)
    out = {"workload": wl["name"], "frame": args.frame, "live_pairs": int(live), "quadrant_waves": 4 * T, "waves_walking_a_list": w[0],
           "staged_batches": w[1], "entries_past_reach_test": w[2], "entries_with_live_pixel": w[3], "entries_contributing": w[4],
           "contributing_lanes": w[5], "live_lanes": w[12],
           "mean_contributing_lanes_per_contributing_entry": round(w[5] / max(1, w[4]), 2),
           "contributing_entries_by_lanes": dict(zip(("1-2", "3-4", "5-8", "9-16", "17-32", "33-64"), w[6:12])),
           "entries_per_live_pair": round(w[2] / max(1, live), 3),
           "frac_reaching_exp": round(w[3] / max(1, w[2]), 4), "frac_reaching_gradient_block": round(w[4] / max(1, w[2]), 4)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
