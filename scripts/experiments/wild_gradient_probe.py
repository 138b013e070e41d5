"""Where this library's gradients and the reference's own backward kernels (compiled for this GPU) part on the wild inputs of
tests/test_reference_hip_gpu.py::wild_training_case, and how far the reference is from ITSELF on a second run (its sums are
formed by atomics in arrival order).  usage: wild_gradient_probe.py [seed] [view z of the 3 % closest splats, e.g. 0.2001]"""
import math, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from autovfx_amd.frame_parallel import settings_for_camera
from diff_gaussian_rasterization import GaussianRasterizer
from oracle import ref_hip
from test_reference_hip_gpu import wild_training_case
dev = torch.device("cuda", 0)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cloud, cam, bg, (w_c, w_d, w_a) = wild_training_case(seed, dev)
if len(sys.argv) > 2:   # pin the closest splats at one view depth
    close = (cloud.means3D[:, 2] + 4.0 > 0.2) & (cloud.means3D[:, 2] + 4.0 < 0.3)
    cloud.means3D[close, 2] = -4.0 + float(sys.argv[2])
P, W, H = cloud.P, cam.image_width, cam.image_height
n_ref, c_ref, d_ref, a_ref, r_ref = ref_hip.forward(cloud, cam, bg)
ref = ref_hip.backward(cloud, cam, bg, n_ref, r_ref, a_ref, w_c, w_d, w_a)
ref2 = ref_hip.backward(cloud, cam, bg, n_ref, r_ref, a_ref, w_c, w_d, w_a)
leaves = {k: getattr(cloud, k).clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
m2d = torch.zeros_like(leaves["means3D"], requires_grad=True)
color, depth, alpha, radii = GaussianRasterizer(settings_for_camera(cam, bg, 3))(leaves["means3D"], m2d, leaves["opacities"], shs=leaves["shs"], scales=leaves["scales"], rotations=leaves["rotations"])
((color * w_c).sum() + (depth * w_d).sum() + (alpha * w_a).sum()).backward()
torch.cuda.synchronize()
print("image", W, H, "rendered", int((radii > 0).sum()))
pairs = {"means3D": leaves["means3D"].grad, "opacity": leaves["opacities"].grad, "sh": leaves["shs"].grad, "scales": leaves["scales"].grad, "rotations": leaves["rotations"].grad, "means2D": m2d.grad}
for k, got in pairs.items():
    want, again = ref[k].reshape(got.shape), ref2[k].reshape(got.shape)
    e = (got - want).abs().reshape(P, -1).amax(1)
    self_e = (again - want).abs().reshape(P, -1).amax(1)
    print(f"{k:10s} max|ref| {float(want.abs().max()):.4g}  ours-ref {float(e.max()):.4g} (id {int(e.argmax())})  ref-ref {float(self_e.max()):.4g} (id {int(self_e.argmax())})")
for k in ("means3D", "scales", "rotations"):
    e = (pairs[k] - ref[k].reshape(pairs[k].shape)).abs().reshape(P, -1).amax(1)
    for i in torch.argsort(e, descending=True)[:3].tolist():
        vz = float(cloud.means3D[i, 2]) + 4.0
        print(f" {k} id {i} err {float(e[i]):.4g}\n    ours {pairs[k][i].tolist()}\n    ref  {ref[k].reshape(pairs[k].shape)[i].tolist()}\n    ref2 {ref2[k].reshape(pairs[k].shape)[i].tolist()}\n    view z {vz:.5f} xy {cloud.means3D[i, :2].tolist()} scales {cloud.scales[i].tolist()} opacity {float(cloud.opacities[i]):.4f} radius {int(radii[i])} |q| {float(cloud.rotations[i].norm()):.3g}")
