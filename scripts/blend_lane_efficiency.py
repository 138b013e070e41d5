"""How much of the blend's per-quadrant walk is spent on pixels that are already finished?  From one C3 frame's
n_contrib (index of each pixel's last contributing list entry) and tile ranges: a quadrant wave walks (about) up to the
largest n_contrib of its 64 pixels; a pixel is useful up to its own.  Prints sum(own) / sum(64 * max), i.e. the lane
efficiency an ideal repacking of live pixels could recover."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autovfx_amd import scenes
from autovfx_amd.cameras import orbit_cameras
from helpers import hip_forward_raw

W, H = 1920, 1080
cloud = scenes.config_c3()
for frame in (10, 210, 410):
    cam = orbit_cameras(800, W, H)[frame]
    out = hip_forward_raw(cloud, cam, debug=False, cull=True)
    nc = np.asarray(out["n_contrib"]).reshape(H, W).astype(np.int64)
    rng = np.asarray(out["ranges"]).reshape(-1, 2).astype(np.int64)
    gx = (W + 15) // 16
    Hp, Wp = (H + 15) // 16 * 16, gx * 16
    pad = np.zeros((Hp, Wp), np.int64); pad[:H, :W] = nc
    q = pad.reshape(Hp // 8, 8, Wp // 8, 8).transpose(0, 2, 1, 3).reshape(Hp // 8, Wp // 8, 64)
    qmax, qsum = q.max(axis=2), q.sum(axis=2)
    tile_len = (rng[:, 1] - rng[:, 0]).reshape(Hp // 16, gx)
    qlen = np.repeat(np.repeat(tile_len, 2, axis=0), 2, axis=1)
    print(f"frame {frame}: mean list length per tile {tile_len.mean():.0f}; mean walk (max n_contrib) per quadrant {qmax.mean():.0f} "
          f"= {100 * qmax.sum() / qlen.sum():.0f}% of the list; useful pixel-entries / (64 x walk) = {qsum.sum() / (64 * qmax.sum()):.3f}")
    # the same with 16-pixel (4x4) groups: what finer-grained termination could reach
    s = pad.reshape(Hp // 4, 4, Wp // 4, 4).transpose(0, 2, 1, 3).reshape(Hp // 4, Wp // 4, 16)
    print(f"           with 4x4 groups: useful / (16 x group walk) = {s.sum() / (16 * s.max(axis=2).sum()):.3f}; "
          f"sum of 4x4 group walks / (4 x quadrant walk) = {s.max(axis=2).sum() / (4 * qmax.sum()):.3f}")
