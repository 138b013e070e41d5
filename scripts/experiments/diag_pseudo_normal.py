import torch, numpy as np, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from autovfx_amd import renderer
from autovfx_amd.cameras import orbit_cameras
import test_raw_gpu as T
DEV="cuda:0"
cam = orbit_cameras(12, 320, 200)[5].to(DEV)
m = T.raw_model(40000, 350)
bg = torch.tensor([0.3,0.1,0.2], device=DEV)
with torch.no_grad():
    got = renderer.render(cam, m, renderer.PipelineParams, bg)
    want = T.reference_shaped_render(cam, m, bg)
for k in T.RENDER_KEYS:
    d = (got[k].float()-want[k].float()).abs()
    print(k, int((d>0).sum()), d.numel(), float(d.max()))
g, w = got["pseudo_normal"], want["pseudo_normal"]
bad = ((g-w).abs()>0).any(-1)
ys, xs = torch.nonzero(bad, as_tuple=True)
print("bad px", int(bad.sum()), "rows range", int(ys.min()), int(ys.max()), "cols", int(xs.min()), int(xs.max()))
depth = got["depth"]
print("bad where depth==0:", int((depth[bad]==0).sum()))
for i in range(5):
    y,x = int(ys[i]), int(xs[i])
    print(y,x, g[y,x].tolist(), w[y,x].tolist(), float(depth[y,x]))
# intermediates in torch
h, wd = 200, 320
from autovfx_amd.cameras import fov2focal
fx, fy = fov2focal(cam.FoVx, wd), fov2focal(cam.FoVy, h)
c2w = cam.view_world_transform
print("c2w", c2w)
directions = renderer.get_ray_directions(h, wd, fx, fy, wd/2, h/2, DEV)
rays_d = directions @ c2w[:3,:3].T
dx, dy = directions[...,0], directions[...,1]
M = c2w[:3,:3]
for j in range(3):
    cand = torch.addcmul(dx*M[j,0], dy, M[j,1])  # not fma-guaranteed
    import itertools
    a0,a1 = dx.double(), dy.double()
    f1 = (a1*M[j,1].double() + (dx*M[j,0]).double()).float()
    c1 = f1 + M[j,2]
    c2 = ((1.0*M[j,2].double()) + f1.double()).float()
    print(j, "match fma-chain:", float((c1==rays_d[...,j]).float().mean()), float((c2==rays_d[...,j]).float().mean()))
    # unfused
    c3 = (dx*M[j,0] + dy*M[j,1]) + M[j,2]
    print(j, "match unfused:", float((c3==rays_d[...,j]).float().mean()))
    # other orders
    f2 = (a0*M[j,0].double() + (dy*M[j,1]).double()).float() + M[j,2]
    print(j, "match fma(a0,m0,a1m1)+m2:", float((f2==rays_d[...,j]).float().mean()))
    f3 = (a1*M[j,1].double() + (a0*M[j,0].double() + M[j,2].double()).float().double()).float()
    print(j, "match fma(a1,m1,fma(a0,m0,m2)):", float((f3==rays_d[...,j]).float().mean()))
    f4 = (a0*M[j,0].double() + (a1*M[j,1].double() + M[j,2].double()).float().double()).float()
    print(j, "match fma(a0,m0,fma(a1,m1,m2)):", float((f4==rays_d[...,j]).float().mean()))
