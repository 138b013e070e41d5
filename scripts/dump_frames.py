#!/usr/bin/env python
"""Dump the 8-bit images of a few C2 frames (what the frame loop's three PNGs hold) for offline work on the PNG encoder, with
the sizes PIL / zlib give them.  Output: <out>/frames.npz + sizes.json."""
import io
import json
import os
import sys
import zlib

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autovfx_amd import renderer, scenes  # noqa: E402
from autovfx_amd.cameras import orbit_cameras  # noqa: E402
from autovfx_amd.frame_io import _frame_to_host, depth2img  # noqa: E402
from autovfx_amd.gaussian_model import GaussianModel  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
dev = torch.device("cuda:0")
c = scenes.config_c2()
model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3).to(dev)
cams = orbit_cameras(200, 960, 540)
bg = torch.zeros(3, device=dev)
arrays, sizes = {}, {}
with torch.no_grad():
    for f in (0, 67, 133):
        res = renderer.render(cams[f].to(dev), model, renderer.PipelineParams, bg)
        rgba8, depth, normal = _frame_to_host(res)
        imgs = {"rgba": rgba8, "depth": depth2img(depth.squeeze(), 3.0), "normal": normal}
        for k, a in imgs.items():
            arrays[f"{k}_{f}"] = a
            b = io.BytesIO()
            Image.fromarray(a).save(b, format="PNG")
            sizes[f"{k}_{f}"] = {"raw": int(a.size), "pil_default": b.getbuffer().nbytes,
                                 "zlib6_unfiltered": len(zlib.compress(a.tobytes(), 6))}
np.savez_compressed(os.path.join(out, "frames.npz"), **arrays)
json.dump(sizes, open(os.path.join(out, "sizes.json"), "w"), indent=1)
print(json.dumps(sizes))
