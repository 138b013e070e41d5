#!/usr/bin/env python
"""How many RENDERED Gaussians (radii > 0) of a training frame receive no gradient at all (never composited: behind an opaque
front, or alpha < 1/255 everywhere)?  For those preprocess_backward_kernel reads 400 bytes of parameters to compute zeros."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
from autovfx_amd.frame_parallel import settings_for_camera
from diff_gaussian_rasterization import GaussianRasterizer

dev = torch.device("cuda", 0)
out = {}
for name, (make, W, H, F) in {"c2": (scenes.config_c2, 960, 540, 200), "c3": (scenes.config_c3, 1920, 1080, 800),
                              "heavy": (scenes.config_heavy, 960, 540, 200)}.items():
    cloud = make().to(dev)
    cam = orbit_cameras(F, W, H)[7].to(dev)
    leaves = [t.clone().requires_grad_(True) for t in (cloud.means3D, cloud.opacities, cloud.shs, cloud.scales, cloud.rotations)]
    m3, op, sh, sc, rot = leaves
    rast = GaussianRasterizer(settings_for_camera(cam, torch.zeros(3, device=dev), 3))
    img, depth, alpha, radii = rast(m3, torch.zeros_like(m3, requires_grad=True), op, shs=sh, scales=sc, rotations=rot)
    ((img - 0.5).abs().mean() + 0.01 * depth.mean()).backward()
    rendered = radii > 0
    untouched = rendered & (op.grad.reshape(-1) == 0) & (sh.grad.reshape(cloud.P, -1) == 0).all(1) & (m3.grad == 0).all(1)
    out[name] = {"P": cloud.P, "rendered": int(rendered.sum()), "rendered_without_any_gradient": int(untouched.sum())}
print(json.dumps(out))
