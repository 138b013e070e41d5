#!/usr/bin/env python
"""Where the frame loop's per-call set-up goes (bench also.c5_loop's scene): object PLYs, DynamicScene construction, the first compose."""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from autovfx_amd import scenes
from autovfx_amd.dynamic_scene import DynamicScene
from autovfx_amd.gaussian_model import GaussianModel
dev = torch.device("cuda:0")
c = scenes.config_c2()
base = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, 3).to(dev)
d = tempfile.mkdtemp(dir="/dev/shm")
paths = []
for k in range(2):
    o = scenes.config_c1(P=60_000, seed=70 + k)
    p = os.path.join(d, f"o{k}.ply")
    GaussianModel.from_activated(o.means3D * 0.25, o.opacities, o.scales * 0.25, o.rotations, o.shs, 3).save_ply(p)
    paths.append(p)
def sync():
    torch.cuda.synchronize()
for rep in range(4):
    sync(); t0 = time.perf_counter()
    objs = {f"o{k}": (GaussianModel(3).load_ply(p, device="cuda:0"), (0, 0, 0)) for k, p in enumerate(paths)}
    sync(); t1 = time.perf_counter()
    for S in (3,):
        scene = DynamicScene(base, objs, device=dev, sh_degree=3, slots=S)
        sync(); t2 = time.perf_counter()
        scene.compose_model([("o0", (0.1, 0.2, 0.3), [[1, 0, 0], [0, 1, 0], [0, 0, 1]], 1.0)], slot=0)
        sync(); t3 = time.perf_counter()
    print(f"rep {rep}: object PLYs {1e3*(t1-t0):.1f} ms, DynamicScene(slots=3) {1e3*(t2-t1):.1f} ms, first compose {1e3*(t3-t2):.1f} ms")
    del scene
