import sys, time, torch, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from autovfx_amd import scenes, _lib
from autovfx_amd.scenes import GaussianCloud
from helpers import settings_for
from diff_gaussian_rasterization import GaussianRasterizer, _C
dev = "cuda:0"
cam = scenes.c1_camera(3840, 2160)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000
g = torch.Generator().manual_seed(1)
means = torch.zeros(P, 3); means[:, :2] = (torch.rand(P, 2, generator=g) - 0.5) * 0.5; means[:, 2] = torch.rand(P, generator=g) * 0.5
rot = torch.zeros(P, 4); rot[:, 0] = 1.0
big = GaussianCloud(means, torch.full((P, 1), 0.5), torch.full((P, 3), 40.0), rot, None, torch.rand(P, 3, generator=g), 0).to(dev)
st = settings_for(cam, dev, (0.0, 0.0, 0.0), 1.0, 0)
outs = []
for mode in ("grad", "nograd"):
    t0 = time.time()
    if mode == "grad":
        m = big.means3D.clone().requires_grad_(True)
        c, d, a, r = GaussianRasterizer(st)(means3D=m, means2D=torch.zeros_like(m), opacities=big.opacities, colors_precomp=big.colors_precomp, scales=big.scales, rotations=big.rotations)
    else:
        with torch.no_grad():
            c, d, a, r = GaussianRasterizer(st)(means3D=big.means3D, means2D=torch.zeros_like(big.means3D), opacities=big.opacities, colors_precomp=big.colors_precomp, scales=big.scales, rotations=big.rotations)
    torch.cuda.synchronize()
    print(mode, "seconds", round(time.time() - t0, 2), "layout", _C.last_layout()["counts"], "alpha min", float(a.min()), "mem GB", round(torch.cuda.max_memory_allocated() / 1e9, 1))
    outs.append((c.detach(), d.detach(), a.detach(), r))
for x, y in zip(outs[0], outs[1]):
    print("equal", bool(torch.equal(x, y)))
