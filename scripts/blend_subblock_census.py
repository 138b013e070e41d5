"""Offline census of the blend's inner loop on one C3 frame (CPU only: the oracle's lists, numpy for the walk).

For a sample of tiles it replays what blend_quadrant_kernel does -- the library's tile culling, the quadrant reach test
(gsr_device.h: splat_reaches_rect), every pixel's front-to-back walk with the reference's stopping rule -- and counts
  * the (entry, 8x8 quadrant) iterations the kernel executes, how many have a live pixel, how many lanes are live;
  * what the same walk would cost if each 4x4 sub-block (16 lanes) or each 8x4 half (32 lanes) of the wave followed its
    OWN reach-filtered list, 64 staged entries at a time (iterations = the longest of the wave's lists per batch).
The second part is the experiment behind profiles/r03_blend_census.md: finer-grained lists would cut the iteration count
by less than a quarter while every iteration would lose its wave-uniform early-outs, so the quadrant form stays.
    python scripts/blend_subblock_census.py [frame] [tiles]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from autovfx_amd import scenes  # noqa: E402
from autovfx_amd.cameras import orbit_cameras  # noqa: E402
from oracle import cpu_oracle  # noqa: E402
from helpers import oracle_kwargs  # noqa: E402

FRAME = int(sys.argv[1]) if len(sys.argv) > 1 else 400
NTILES = int(sys.argv[2]) if len(sys.argv) > 2 else 160
W, H = 1920, 1080; gx = 120; gy = 68
t0 = time.time()
z = cpu_oracle.forward(intermediates=True, **oracle_kwargs(scenes.config_c3(), orbit_cameras(800, W, H)[FRAME]))
print(f"C3 frame {FRAME}: oracle forward in {time.time() - t0:.1f} s, num_rendered {z['num_rendered']}")
pl, rng, m2, co = z["point_list"], z["ranges"], z["means2D"].astype(np.float64), z["conic_opacity"].astype(np.float64)

def reach(A, B, C, o, cx, cy, x0, y0, w, h):
    """splat_reaches_rect (gsr_device.h), vectorised over entries"""
    skip = -np.log(255.0 * o) - 1e-4
    budget = -skip - 1e-4
    ok_reg = (A > 0) & (C > 0) & (B * B < 0.99 * A * C)
    x_lo, x_hi = cx - (x0 + w - 1), cx - x0
    y_lo, y_hi = cy - (y0 + h - 1), cy - y0
    inside = (x_lo <= 0) & (x_hi >= 0) & (y_lo <= 0) & (y_hi >= 0)
    nb_c, nb_a = -B / C, -B / A
    q = lambda dx, dy: 0.5 * (A * dx * dx + C * dy * dy) + B * dx * dy
    qmin = q(x_lo, np.minimum(y_hi, np.maximum(y_lo, nb_c * x_lo)))
    qmin = np.minimum(qmin, q(x_hi, np.minimum(y_hi, np.maximum(y_lo, nb_c * x_hi))))
    qmin = np.minimum(qmin, q(np.minimum(x_hi, np.maximum(x_lo, nb_a * y_lo)), y_lo))
    qmin = np.minimum(qmin, q(np.minimum(x_hi, np.maximum(x_lo, nb_a * y_hi)), y_hi))
    return (~ok_reg) | inside | ~(qmin * 0.998 - 1e-3 > budget)

rs = np.random.default_rng(0)
tiles = rs.choice(gx * gy, NTILES, replace=False)
tot = dict(entries=0, tile_reach=0, it_quad=0, it_quad_live=0, live_lanes=0, it_sub=0, it_sub_sum=0, it_half=0, it_sub_nobatch=0, pairs_sub=0, stageB_quad=0, stageB_sub=0)
for t in tiles:
    a, b = rng[t]
    ids = pl[a:b]
    L = len(ids)
    if L == 0: continue
    ty, tx = divmod(t, gx)
    A, B, C, o = co[ids].T
    cx, cy = m2[ids].T
    # tile-level culling as the library does (live tiles): entries the library would list
    keep = reach(A, B, C, o, cx, cy, tx * 16, ty * 16, 16, 16)
    ids, A, B, C, o, cx, cy = ids[keep], A[keep], B[keep], C[keep], o[keep], cx[keep], cy[keep]
    L = len(ids)
    tot["entries"] += int(keep.size); tot["tile_reach"] += L
    ys, xs = np.mgrid[ty * 16:ty * 16 + 16, tx * 16:tx * 16 + 16]
    dx = cx[:, None, None] - xs[None]; dy = cy[:, None, None] - ys[None]
    power = -0.5 * (A[:, None, None] * dx * dx + C[:, None, None] * dy * dy) - B[:, None, None] * dx * dy
    alpha = np.minimum(0.99, o[:, None, None] * np.exp(np.minimum(power, 0)))
    live = (power <= 0) & (alpha >= 1 / 255)            # would blend if not done
    inside = (xs < W) & (ys < H)
    # sequential T: done when test_T < 1e-4
    T = np.ones((16, 16)); done_at = np.full((16, 16), L, np.int64)   # index of the entry at which the pixel stops (L = never)
    done = ~inside
    for e in range(L):
        l = live[e] & ~done
        test_T = T * (1 - alpha[e])
        stop = l & (test_T < 1e-4)
        done_at[stop] = e
        done |= stop
        upd = l & ~stop
        T = np.where(upd, test_T, T)
        if done.all(): break
    done_at[~inside] = -1
    # coarser work items (two or four pixels per lane): a 16x8 / 8x16 half tile or the whole tile per wave, one reach-filtered list
    for name, rects in (("h16x8", [(0, 0, 16, 8), (0, 8, 16, 8)]), ("v8x16", [(0, 0, 8, 16), (8, 0, 8, 16)]), ("full16", [(0, 0, 16, 16)])):
        for (ox, oy, w, h) in rects:
            sl2 = (slice(oy, oy + h), slice(ox, ox + w))
            if not inside[sl2].any(): continue
            n_h = min(L, done_at[sl2].max() + 1)
            rh = reach(A[:n_h], B[:n_h], C[:n_h], o[:n_h], cx[:n_h], cy[:n_h], tx * 16 + ox, ty * 16 + oy, w, h)
            tot["it_" + name] = tot.get("it_" + name, 0) + int(rh.sum())
    for qy in range(2):
        for qx in range(2):
            sl = (slice(8 * qy, 8 * qy + 8), slice(8 * qx, 8 * qx + 8))
            if not inside[sl].any(): continue
            stop_q = done_at[sl].max()          # the quadrant processes entries 0..stop_q (inclusive) (or all if L)
            n_q = min(L, stop_q + 1)
            rq = reach(A[:n_q], B[:n_q], C[:n_q], o[:n_q], cx[:n_q], cy[:n_q], tx * 16 + 8 * qx, ty * 16 + 8 * qy, 8, 8)
            tot["it_quad"] += int(rq.sum())
            # live lanes at the time (pixel not yet done and live)
            ee = np.arange(n_q)[:, None, None]
            lv = live[:n_q][:, sl[0], sl[1]] & (ee <= done_at[sl][None])
            anyl = lv.reshape(n_q, -1).any(1) & rq
            tot["it_quad_live"] += int(anyl.sum()); tot["live_lanes"] += int(lv[rq].sum())
            # 4x4 sub-blocks: own reach lists, own stop
            rsub = np.zeros((4, n_q), bool)
            g = 0
            for sy in range(2):
                for sx in range(2):
                    ssl = (slice(8 * qy + 4 * sy, 8 * qy + 4 * sy + 4), slice(8 * qx + 4 * sx, 8 * qx + 4 * sx + 4))
                    stop_g = done_at[ssl].max()
                    n_g = min(n_q, stop_g + 1)
                    r = reach(A[:n_g], B[:n_g], C[:n_g], o[:n_g], cx[:n_g], cy[:n_g], tx * 16 + 8 * qx + 4 * sx, ty * 16 + 8 * qy + 4 * sy, 4, 4)
                    rsub[g, :n_g] = r & rq[:n_g]
                    g += 1
            tot["pairs_sub"] += int(rsub.sum())
            # batched by 64 list entries (as the kernel stages them): iterations per batch = max over groups
            for b0 in range(0, n_q, 64):
                c = rsub[:, b0:b0 + 64].sum(1)
                tot["it_sub"] += int(c.max())
                # halves (8x4): groups (0,1) upper, (2,3) lower
                up = (rsub[0, b0:b0 + 64] | rsub[1, b0:b0 + 64]).sum(); lo = (rsub[2, b0:b0 + 64] | rsub[3, b0:b0 + 64]).sum()
                tot["it_half"] += int(max(up, lo))
            tot["it_sub_nobatch"] += int(rsub.sum(1).max())
print({k: v for k, v in tot.items()})
print("reference list entries / tile-culled", tot["entries"], tot["tile_reach"])
print("quadrant iterations (processed entries):", tot["it_quad"], " with a live pixel: %.1f%%" % (100 * tot["it_quad_live"] / tot["it_quad"]), " live lanes per processed entry: %.1f" % (tot["live_lanes"] / tot["it_quad"]))
print("4x4 sub-block streams: iterations %.1f%% of now (batched by 64), %.1f%% unbatched; (sub-block, entry) pairs per quadrant iteration %.2f" % (100 * tot["it_sub"] / tot["it_quad"], 100 * tot["it_sub_nobatch"] / tot["it_quad"], tot["pairs_sub"] / tot["it_quad"]))
print("8x4 halves: iterations %.1f%% of now" % (100 * tot["it_half"] / tot["it_quad"]))
for name, what in (("h16x8", "16x8 halves, 2 pixels per lane"), ("v8x16", "8x16 halves, 2 pixels per lane"), ("full16", "whole tile, 4 pixels per lane")):
    print("%s: iterations %.1f%% of now" % (what, 100 * tot["it_" + name] / tot["it_quad"]))
