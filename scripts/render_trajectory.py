#!/usr/bin/env python
"""Render a camera trajectory of a 3DGS scene to disk: what AutoVFX's ``scene_representation.render_from_3DGS``
(`scene_representation.py:355-438`) does for the static background, through this repository's drop-in renderer.

    python scripts/render_trajectory.py --ply point_cloud.ply --trajectory custom_camera_path/traj.json --out out/
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/render_trajectory.py ...   # frames sharded

Per frame it writes images/<i>.png (RGBA8, save_image rounding), depth/<i>.npy (fp32) and normal/<i>.png, the three
files the reference's frame loop writes and `blender/blend_all.py` consumes.  Rank r renders frames r, r+N, ...; every
rank writes its own frames (no gather is needed when the output is files).  `--synthetic P` renders the synthetic C2
cloud with P Gaussians instead of a PLY; `--orbit N WxH` an N-view orbit instead of a trajectory file.

Moving inserted objects (`scene_representation.py:357-372`): `--object ID=object_gaussians.ply@cx,cy,cz` (repeatable; the
object's Gaussians and its initial centre, what `get_center_of_mesh_2` returns for its mesh) and `--rigid-body-json FILE` with the
reference's `rb_transform_info` structure, `{ID: {"001": {"pos": [x, y, z], "rot": [[...], [...], [...]], "scale": s}, ...}}`
(frame keys are 1-based, three digits; an object without an entry for a frame is absent from it).  The scene then lives in
resident buffers and each frame places its objects with one kernel each (`autovfx_amd/dynamic_scene.py`).
"""
import argparse
import json
import os
import sys
import time

import os as _os
_os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # before the HIP runtime starts: RCCL's peer set-up needs dmabuf IPC here
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ply", help="vanilla 3DGS point_cloud.ply (read with --sh-degree)")
    ap.add_argument("--checkpoint", help="SuGaR .pt checkpoint or 3DGS .ply, dispatched like scene_representation.load_scene "
                                         "(--sh-degree + 1 plays the role of its max_sh_degree)")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--trajectory")
    ap.add_argument("--orbit", nargs=2, metavar=("N", "WxH"))
    ap.add_argument("--downscale", type=float, default=1.0)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--white-background", action="store_true")
    ap.add_argument("--object", action="append", default=[], metavar="ID=PLY@cx,cy,cz",
                    help="an inserted object: its Gaussians (PLY, read with --sh-degree) and its initial centre")
    ap.add_argument("--rigid-body-json", help="per-frame rigid-body transforms of the objects (rb_transform_info)")
    ap.add_argument("--full-sh-on-placed-frames", action="store_true",
                    help="render frames with placed objects at the scene's SH degree.  Default: degree 0 on such frames, as the "
                         "reference does (its merged model is a fresh GaussianModel whose active_sh_degree is never raised: "
                         "gaussians_utils.py:71-82)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--deflate", action="store_true",
                    help="compress the PNGs on a pool of host threads (zlib level 3) instead of building stored-deflate file images on the GPU")
    ap.add_argument("--writer-threads", type=int, default=4, help="host threads that write() the GPU-built file images")
    ap.add_argument("--streams", type=int, default=3,
                    help="frames in flight on as many HIP streams, fed by this one host thread through render_begin / finish (1: one blocking render() per frame)")
    args = ap.parse_args()

    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    if not torch.cuda.is_available():
        raise SystemExit("needs a HIP device: the render path has no CPU fallback")
    from autovfx_amd.frame_parallel import local_device
    dev = local_device()                        # cuda:LOCAL_RANK
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from autovfx_amd import cameras, frame_io, renderer, scenes
    from autovfx_amd.frame_parallel import shard_frames
    from autovfx_amd.gaussian_model import GaussianModel

    if args.checkpoint:
        from autovfx_amd.gaussian_model import load_scene
        model = load_scene(args.checkpoint, args.sh_degree + 1, device=str(dev))
    elif args.ply:
        model = GaussianModel(args.sh_degree).load_ply(args.ply, device=str(dev))
        model.active_sh_degree = model.max_sh_degree
    else:
        c = scenes.config_c2(P=args.synthetic or 200_000)
        model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, c.sh_degree).to(dev)
    if args.trajectory:
        cams = cameras.cameras_from_trajectory(args.trajectory, args.downscale)
    else:
        n, wh = args.orbit or ("8", "960x540")
        w, h = (int(v) for v in wh.lower().split("x"))
        cams = cameras.orbit_cameras(int(n), w, h)
    bg = torch.tensor([1.0, 1.0, 1.0] if args.white_background else [0.0, 0.0, 0.0], device=dev)

    scene, transforms = None, {}
    if args.object:
        from autovfx_amd.dynamic_scene import DynamicScene
        objects = {}
        for spec in args.object:
            oid, rest = spec.split("=", 1)
            path, centre = rest.rsplit("@", 1)
            om = GaussianModel(args.sh_degree).load_ply(path, device=str(dev))
            objects[oid] = (om, [float(v) for v in centre.split(",")])
        if args.rigid_body_json:
            with open(args.rigid_body_json) as f:
                transforms = json.load(f)
        scene = DynamicScene(model, objects, device=dev, sh_degree=model.active_sh_degree, slots=max(1, args.streams),
                             placed_sh_degree=None if args.full_sh_on_placed_frames else 0)

    def frame_model(i, slot=0):
        if scene is None:
            return model
        key = "{0:03d}".format(i + 1)   # frame index starts from 001 (scene_representation.py:362)
        placed = [(oid, t[key]["pos"], t[key]["rot"], float(t[key]["scale"])) for oid, t in transforms.items() if key in t]
        return scene.compose_model(placed, slot=slot)   # (a frame in flight keeps its own copy of the scene buffers)

    mine = shard_frames(len(cams), rank, world)
    cams_dev = dict(zip(mine, cameras.Camera.batch_to([cams[i] for i in mine], dev)))   # this rank's cameras, uploaded in five copies
    t0 = time.perf_counter()
    # file images built on the GPU, host threads only write(); --deflate: the frame crosses as pixels and a pool of host threads
    # compresses it (zlib level 3)
    S = max(1, args.streams)
    from autovfx_amd import frame_loop
    if args.deflate:   # host zlib pool instead of GPU-built file images
        frame_loop._make_writer = lambda out_dir, threads, slots: frame_io.FrameWriter(out_dir)
    names = [c.image_name or f"{i:05d}" for i, c in enumerate(cams)]
    views = [cams_dev.get(i) for i in range(len(cams))]
    with torch.no_grad():
        # S frames in flight from this one thread (autovfx_amd/frame_loop.py: the loop behind SceneRepresentation.render_from_3DGS)
        frame_loop.render_frames(views, names, frame_model, args.out, renderer.PipelineParams, bg, frame_ids=mine, streams=S,
                                 writer_threads=args.writer_threads)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"rank": rank, "frames": len(mine), "seconds": round(dt, 3),
                      "frames_per_s_including_png_encode": round(len(mine) / max(dt, 1e-9), 2),
                      "writer": "host zlib pool" if args.deflate else f"GPU file images, {args.writer_threads} writer threads",
                      "streams": S}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
