"""Debug: where the product and the reference's kernels (compiled for this GPU) differ for non-finite inputs."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from autovfx_amd import scenes, _lib
from autovfx_amd.scenes import GaussianCloud
from autovfx_amd.frame_parallel import rasterize
from oracle import ref_hip
dev = torch.device("cuda", 0)
cloud, cam0 = scenes.config_c1(P=4000, seed=17), scenes.c1_camera(160, 96)
for field, poison in (("scales", float("nan")), ("scales", float("inf")), ("rotations", float("nan")), ("shs", float("nan")), ("opacities", -float("inf"))):
    g = torch.Generator().manual_seed(3)
    bad = torch.randperm(cloud.P, generator=g)[:40]
    dirty = GaussianCloud(cloud.means3D.clone(), cloud.opacities.clone().reshape(cloud.P, 1), cloud.scales.clone(), cloud.rotations.clone(), cloud.shs.clone(), None, 3)
    flat = getattr(dirty, field).reshape(cloud.P, -1)
    flat[bad, torch.randint(0, flat.shape[1], (40,), generator=g)] = poison
    dirty, cam = dirty.to(dev), cam0.to(dev)
    bg = torch.tensor([0.1, 0.0, 0.2], device=dev)
    n_ref, c_ref, d_ref, a_ref, r_ref = ref_hip.forward(dirty, cam, bg)
    for cull in (1, 0):
        _lib.set_option(_lib.OPT_TILE_CULL, cull)
        with torch.no_grad():
            color, depth, alpha, radii = rasterize(dirty, cam, bg)
        tag = lambda t: torch.nan_to_num(t, nan=7e8, posinf=8e8, neginf=-8e8)
        diff = ((tag(color) - tag(c_ref)).abs() > 1e-4).any(0)
        ys, xs = torch.nonzero(diff, as_tuple=True)
        tiles = sorted(set((int(y) // 16, int(x) // 16) for y, x in zip(ys.tolist(), xs.tolist())))
        print(field, poison, "cull", cull, "px differ", int(diff.sum()), "tiles", tiles[:8], "nan px ours/ref", int(torch.isnan(color).any(0).sum()), int(torch.isnan(c_ref).any(0).sum()),
              "poisoned radii ref", r_ref[bad.to(dev)].tolist()[:6])
        if len(ys):
            y, x = int(ys[0]), int(xs[0])
            print("   first px", (y, x), "ours", color[:, y, x].tolist(), float(alpha[0, y, x]), "ref", c_ref[:, y, x].tolist(), float(a_ref[0, y, x]))
    _lib.set_option(_lib.OPT_TILE_CULL, 1)
