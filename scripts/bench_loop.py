#!/usr/bin/env python
"""One leg of bench.py's `also` object on its own: `also.c5_loop` (the whole frame loop of an edited scene to a tmpfs, the
reference-shaped loop beside it) and, with --blend, `also.c5_blend_frames`.  Prints one JSON line per leg."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--reference-frames", type=int, default=10)
    ap.add_argument("--legs", default="c5_loop")
    args = ap.parse_args()
    # diagnosis knobs (not product paths): GSR_LOOP_DIAG=nowrite -- the files are built and copied to the host, nothing is written;
    # GSR_LOOP_DIAG=nofiles -- frames are rendered and dropped
    diag = os.environ.get("GSR_LOOP_DIAG", "")
    if diag:
        from autovfx_amd import frame_io, frame_loop

        class _Drop:
            def __enter__(self):
                return self

            def __exit__(self, *a):
                return False

            def submit(self, name, result):
                pass

        if diag == "nofiles":
            frame_loop._make_writer = lambda out_dir, threads, slots: _Drop()
        elif diag == "nowrite":
            def _no_write(slot, paths, off, lengths_at=None):
                slot["event"].synchronize()
                for p_ in paths.values():
                    open(p_, "wb").close()
                return paths
            frame_io.GpuFrameWriter._write = staticmethod(_no_write)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    sys.modules.setdefault("bench", bench)
    for leg in args.legs.split(","):
        if leg == "c5_loop":
            out = bench.frame_loop_bench(dev, frames=args.frames, reference_frames=args.reference_frames)
        elif leg == "c5_blend_frames":
            out = bench.blend_frames_bench(dev, frames=args.frames)
        else:
            raise SystemExit("unknown leg " + leg)
        print(json.dumps({leg: out}), flush=True)
