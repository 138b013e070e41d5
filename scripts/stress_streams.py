"""Race hunt: the same frames through 1 and through S in {2,3,4,5,7} HIP streams, both stream drivers, at the op
boundary (rasterize) on two workloads; every frame's RGBA8 + depth must hash the same as the serial run's."""
import hashlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
from autovfx_amd.frame_parallel import render_shard

dev = torch.device("cuda", 0)
cases = [("c2", scenes.config_c2(), 960, 540, 240), ("c3", scenes.config_c3(), 1920, 1080, 96),
         ("heavy", scenes.config_heavy(), 960, 540, 120)]   # (C3 and heavy are cut into depth slabs, C2 is not)
REPS = int(os.environ.get("STRESS_REPS", "2"))
bg = torch.zeros(3, device=dev)
bad = 0
for name, cloud, W, H, n in cases:
    cloud = cloud.to(dev)
    cams = [c.to(dev) for c in orbit_cameras(800, W, H)[:n]]
    def run(streams, driver):
        t0 = time.perf_counter()
        out = render_shard(cloud, cams, list(range(n)), bg, keep_depth=True, streams=streams, driver=driver)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        hs = [hashlib.sha256(out["rgba8"][i].cpu().numpy().tobytes() + out["depth"][i].cpu().numpy().tobytes()).hexdigest() for i in range(n)]
        return hs, dt
    ref, t1 = run(1, "auto")
    for S in (2, 3, 4, 5, 7):
        for driver in ("pipelined", "threads"):
            for rep in range(REPS):
                hs, dt = run(S, driver)
                diff = sum(a != b for a, b in zip(ref, hs))
                bad += diff
                print(f"{name} S={S} {driver:9s} rep{rep}: {n / dt:7.0f} fps, frames differing from serial: {diff}", flush=True)
print("TOTAL differing frames:", bad)
sys.exit(1 if bad else 0)
