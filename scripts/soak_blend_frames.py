#!/usr/bin/env python
"""Soak of autovfx_amd.compositor.blend_frames: N frames drawn from 4 distinct sets of layers, several pool sizes; every written frame
must be byte-identical to the first frame of its set (races in the workers' staging / page-locked buffers, the batched launches or the
GPU's zlib decoder would show as a differing file).  Prints one JSON line."""
import argparse
import hashlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from autovfx_amd import compositor  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=1200)
ap.add_argument("--threads", default="3,16,32")
args = ap.parse_args()
dev = torch.device("cuda", 0)
root = tempfile.mkdtemp(prefix="gsr_soak_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
out = {"frames": args.frames, "runs": []}
try:
    results, cfg = bench._synthetic_blender_tree(root, 960, 540, args.frames, distinct=4)
    want = None
    for threads in [int(t) for t in args.threads.split(",")]:
        os.environ["AUTOVFX_AMD_BLEND_DECODERS"] = str(threads)
        shutil.rmtree(os.path.join(results, "frames"), ignore_errors=True)
        t0 = time.perf_counter()
        paths = compositor.blend_frames(results, cfg, device=dev, write_video=False)
        dt = time.perf_counter() - t0
        digests = [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in paths]
        if want is None:
            want = digests[:4]
        wrong = [i for i, d in enumerate(digests) if d != want[i % 4]]
        out["runs"].append({"threads": threads, "frames_per_s": round(len(paths) / dt, 1), "differing_frames": wrong[:10], "n_differing": len(wrong)})
    out["ok"] = all(r["n_differing"] == 0 for r in out["runs"]) and len(set(want)) == 4
finally:
    shutil.rmtree(root, ignore_errors=True)
print(json.dumps(out))
sys.exit(0 if out.get("ok") else 1)
