"""cProfile of the per-frame host path (rasterize + pack) on a bench workload: where does the Python time go?"""
import cProfile, pstats, sys, os, io
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
from autovfx_amd.frame_parallel import rasterize, pack_rgba8
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = {"c2": (scenes.config_c2, 960, 540, 200), "c3": (scenes.config_c3, 1920, 1080, 800)}[wl]
dev = torch.device("cuda", 0)
cloud = cfg[0]().to(dev)
cams = [c.to(dev) for c in orbit_cameras(cfg[3], cfg[1], cfg[2])[:64]]
bg = torch.zeros(3, device=dev)
out = torch.empty((4, cfg[2], cfg[1]), dtype=torch.uint8, device=dev)
def frames(n):
    with torch.no_grad():
        for i in range(n):
            c, d, a, r = rasterize(cloud, cams[i % 64], bg)
            pack_rgba8(c, a, out=out)
frames(10); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); frames(300); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
