"""Soak test: render N frames of a workload serially and with 3 streams -- one host thread with split calls, then one
blocking host thread per stream -- through the full render() boundary (memoised activations, fused kernels, the
folded normal pass) and require bit-identical frames."""
import hashlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autovfx_amd import renderer, scenes
from autovfx_amd.cameras import orbit_cameras
from autovfx_amd.frame_parallel import render_shard
from autovfx_amd.gaussian_model import GaussianModel

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda", 0)
c = scenes.config_c2()
model = GaussianModel.from_activated(c.means3D, c.opacities, c.scales, c.rotations, c.shs, c.sh_degree).to(dev)
cams = [k.to(dev) for k in orbit_cameras(200, 960, 540)[:n]]
bg = torch.zeros(3, device=dev)

def render_fn(_cloud, cam, bg_):
    o = renderer.render(cam, model, renderer.PipelineParams, bg_)
    return o["render"][:3], torch.cat((o["depth"][None], o["normal"].permute(2, 0, 1), o["pseudo_normal"].permute(2, 0, 1)), 0), o["render"][3:4], o["radii"]

class Pending:
    def __init__(self, p):
        self.p = p
    def finish(self):
        o = self.p.finish()
        return o["render"][:3], torch.cat((o["depth"][None], o["normal"].permute(2, 0, 1), o["pseudo_normal"].permute(2, 0, 1)), 0), o["render"][3:4], o["radii"]

def begin_fn(_cloud, cam, bg_):
    return Pending(renderer.render_begin(cam, model, renderer.PipelineParams, bg_))

cloud = c.to(dev)
def run(streams, driver="threads"):
    t0 = time.perf_counter()
    out = render_shard(cloud, cams, list(range(n)), bg, keep_depth=True, render_fn=render_fn, streams=streams,
                       driver=driver, begin_fn=begin_fn)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    h = hashlib.sha256(out["rgba8"].cpu().numpy().tobytes() + out["depth"].cpu().numpy().tobytes()).hexdigest()
    return h, dt
h1, t1 = run(1)
for rep in range(3):
    h3, t3 = run(3, "pipelined")
    h4, t4 = run(3, "threads")
    print(f"serial {n / t1:.0f} fps, 3 streams pipelined {n / t3:.0f} fps, threads {n / t4:.0f} fps, "
          f"identical: {h1 == h3 == h4}")
    assert h1 == h3 == h4
