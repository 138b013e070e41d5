"""Diagnostic: what the pair expansion of each depth slab of an inference call has to walk.

    python scripts/diag_expand.py [--workload c3] [--frame 0]

Reads the scratch of one inference call back (slab table, per-position pair offsets) and prints, per slab, how the
pairs are spread over the depth-ordered splats: splats with / without live pairs, and how many splats one 4096-pair
workgroup of expand_kernel spans (it parks their offsets in LDS 2048 at a time).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--frame", type=int, default=0)
    args = ap.parse_args()
    from autovfx_amd import _lib, scenes
    from autovfx_amd.cameras import orbit_cameras
    from diff_gaussian_rasterization import _C
    import bench
    wl = bench.WORKLOADS[args.workload]
    W, H = wl["width"], wl["height"]
    cloud = getattr(scenes, wl["cfg"])().to("cuda")
    cam = orbit_cameras(wl["frames"], W, H)[args.frame].to("cuda")
    bg = torch.zeros(3, device="cuda")
    e = torch.Tensor([])
    _C.set_geometry_cache(False)
    P = cloud.P
    for rep in range(2):
        n, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
            bg, cloud.means3D, e, cloud.opacities, cloud.scales, cloud.rotations, 1.0, e, cam.world_view_transform,
            cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, cloud.shs, cloud.sh_degree, cam.camera_center,
            False, False, inference=True)
    torch.cuda.synchronize()
    lay = _C.last_layout()
    g = lay["geom"]
    shift = g["raster"] - 256                      # the header occupies the first 256 bytes of the aligned base
    hdr = geom[shift:shift + 256].cpu().numpy()
    counts = hdr[8:24].view(np.uint32)
    off = hdr[32:96].view(np.uint64)
    S = int(counts[2])
    slabs = geom[shift + int(off[5]): shift + int(off[5]) + 16 * 8].view(torch.int32).cpu().numpy().view(np.uint32).reshape(8, 4)
    po_at = g["point_offsets"]
    so_at = shift + ((po_at - shift + 4 * P + 255) & ~255)
    po = geom[po_at:po_at + 4 * P].view(torch.int32).cpu().numpy().view(np.uint32).astype(np.int64)
    so = geom[so_at:so_at + 4 * P].view(torch.int32).cpu().numpy().view(np.uint32).astype(np.int64)
    print(f"P {P}  num_rendered {n}  slabs {S}  slab_pairs {lay['slab_pairs']}")
    for s in range(S):
        first, end, pairs = (int(v) for v in slabs[s][:3])
        incl = po[first:end] if s == 0 else so[first:end]
        cnt = np.diff(np.concatenate(([0], incl)))
        emit = int((cnt > 0).sum())
        print(f"slab {s}: positions [{first}, {end}) = {end - first} splats, pairs {pairs} (offsets end {int(incl[-1]) if len(incl) else 0}), "
              f"emitting {emit} ({100.0 * emit / max(1, end - first):.1f} %)")
        if pairs == 0:
            continue
        starts = np.arange(0, pairs, 4096)
        ends = np.minimum(starts + 4096, pairs)
        s0 = np.searchsorted(incl, starts, side="right")      # first splat whose inclusive offset exceeds p_begin
        s1 = np.searchsorted(incl, ends, side="left")         # splat that owns the chunk's last pair
        span = s1 - s0 + 1
        batches = (span + 2047) // 2048
        print(f"   workgroups {len(starts)}: splats spanned mean {span.mean():.0f} p50 {np.median(span):.0f} p99 {np.percentile(span, 99):.0f} "
              f"max {span.max()}  -> LDS batches mean {batches.mean():.2f} max {batches.max()}")
        zr = np.flatnonzero(cnt > 0)
        if len(zr) > 1:
            gaps = np.diff(zr) - 1
            print(f"   runs of splats without pairs between emitting ones: mean {gaps.mean():.1f} p99 {np.percentile(gaps, 99):.0f} max {gaps.max()}")
    _lib.set_stage_timing(True)
    for rep in range(4):
        _C.rasterize_gaussians(bg, cloud.means3D, e, cloud.opacities, cloud.scales, cloud.rotations, 1.0, e,
                               cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, H, W, cloud.shs,
                               cloud.sh_degree, cam.camera_center, False, False, inference=True)
    torch.cuda.synchronize()
    print("stage ms:", {k: round(v, 4) for k, v in _lib.stage_times_ms().items()})


if __name__ == "__main__":
    main()
