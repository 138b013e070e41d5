"""Vectorised PyTorch-CPU splat path: a second, independently written restatement.

TEST INFRASTRUCTURE ONLY (see oracle/gsr_oracle.c for the rule).  The reference has no CPU splat
path of its own (its ``convert_SHs_python`` / ``compute_cov3D_python`` switches still rasterize
in CUDA, ``gaussian_renderer/__init__.py:127-144``); this file is the "PyTorch CPU reference
splat" of BASELINE.json configs[0]: same semantics as SURVEY.md appendix A.3, written with
tensor ops instead of loops, in fp32 or fp64.  It is used to cross-check the C oracle (different
code, different summation order => agreement to ~1e-6, not bit-exact) and is timed as a
secondary CPU baseline.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

TILE = 16
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435)


def _sh_rgb(deg: int, dirs: torch.Tensor, sh: torch.Tensor) -> torch.Tensor:
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    out = C0 * sh[:, 0]
    if deg > 0:
        out = out - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        out = (out + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
               + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        out = (out + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
               + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
               + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
               + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return torch.clamp_min(out + 0.5, 0.0)


def _quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    r, x, y, z = q.unbind(1)
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), dim=1)
    return R.view(-1, 3, 3)  # standard rotation matrix, rows as written


def preprocess(*, means3D, opacities, width, height, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
               sh_degree=0, scale_modifier=1.0, shs=None, colors_precomp=None, scales=None, rotations=None,
               cov3D_precomp=None, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    f = lambda t: None if t is None else torch.as_tensor(t).detach().cpu().to(dtype)
    p, op = f(means3D), f(opacities).reshape(-1)
    V, PM, cam = f(viewmatrix), f(projmatrix), f(campos)
    P = p.shape[0]
    W, H = int(width), int(height)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ones = torch.ones(P, 1, dtype=dtype)
    ph = torch.cat((p, ones), 1) @ PM          # row-vector times transposed matrix
    pv = (torch.cat((p, ones), 1) @ V)[:, :3]
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    vis = pv[:, 2] > 0.2

    if cov3D_precomp is not None:
        c = f(cov3D_precomp)
        Sig = torch.stack((c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]), 1).view(-1, 3, 3)
    else:
        R = _quat_to_rot(f(rotations))
        S = torch.diag_embed(f(scales) * scale_modifier)
        RS = R @ S
        Sig = RS @ RS.transpose(1, 2)          # R S S^T R^T

    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    tz = pv[:, 2]
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tx = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    J = torch.zeros(P, 2, 3, dtype=dtype)
    J[:, 0, 0] = fx / tz
    J[:, 0, 2] = -fx * tx / (tz * tz)
    J[:, 1, 1] = fy / tz
    J[:, 1, 2] = -fy * ty / (tz * tz)
    Wm = V[:3, :3].t()                         # w2c rotation (V is stored transposed)
    JW = J @ Wm
    cov = JW @ Sig @ JW.transpose(1, 2)
    a, b, c_ = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c_ - b * b
    vis = vis & (det != 0)
    det_inv = 1.0 / det
    conic = torch.stack((c_ * det_inv, -b * det_inv, a * det_inv), 1)
    mid = 0.5 * (a + c_)
    disc = torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(torch.maximum(mid + disc, mid - disc)))
    pix = torch.stack((((ndc[:, 0].double() + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1].double() + 1.0) * H - 1.0) * 0.5), 1).to(dtype)
    ri = torch.nan_to_num(radius, nan=0.0).clamp(-2**31, 2**31 - 1).to(torch.int64)
    rf = ri.to(dtype)
    x0 = torch.trunc((pix[:, 0] - rf) / TILE).clamp(0, gx).long()
    y0 = torch.trunc((pix[:, 1] - rf) / TILE).clamp(0, gy).long()
    x1 = torch.trunc((pix[:, 0] + rf + TILE - 1) / TILE).clamp(0, gx).long()
    y1 = torch.trunc((pix[:, 1] + rf + TILE - 1) / TILE).clamp(0, gy).long()
    area = (x1 - x0) * (y1 - y0)
    vis = vis & (area > 0)
    if colors_precomp is not None:
        rgb = f(colors_precomp)
    else:
        d = p - cam[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = _sh_rgb(int(sh_degree), d, f(shs))
    z = torch.zeros((), dtype=dtype)
    return {"visible": vis, "radii": torch.where(vis, ri, 0).to(torch.int32), "means2D": pix, "depths": pv[:, 2],
            "conic_opacity": torch.cat((conic, op[:, None]), 1), "rgb": rgb,
            "tiles_touched": torch.where(vis, area, 0), "rect": torch.stack((x0, y0, x1, y1), 1), "zero": z}


def forward(*, bg, chunk: int = 4096, **kw) -> Dict[str, torch.Tensor]:
    """Full splat; same keyword interface as ``oracle.cpu_oracle.forward``."""
    dtype = kw.get("dtype", torch.float32)
    W, H = int(kw["width"]), int(kw["height"])
    P = int(torch.as_tensor(kw["means3D"]).shape[0])
    color = torch.zeros(3, H, W, dtype=dtype)
    depth = torch.zeros(1, H, W, dtype=dtype)
    alpha = torch.zeros(1, H, W, dtype=dtype)
    if P == 0:
        return {"color": color, "depth": depth, "alpha": alpha, "radii": torch.zeros(0, dtype=torch.int32), "num_rendered": 0}
    g = preprocess(**kw)
    bgv = torch.as_tensor(bg).detach().cpu().to(dtype)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ids = torch.nonzero(g["visible"]).squeeze(1)
    rect = g["rect"][ids]
    nx, ny = rect[:, 2] - rect[:, 0], rect[:, 3] - rect[:, 1]
    cnt = nx * ny
    D = int(cnt.sum())
    owner = torch.repeat_interleave(torch.arange(ids.numel()), cnt)
    first = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(D) - first[owner]
    tile = (rect[owner, 1] + local // nx[owner]) * gx + rect[owner, 0] + local % nx[owner]
    gid = ids[owner]
    # stable order: tile, then depth (as float32 bit pattern order == numeric order for z > 0.2), then id
    d32 = g["depths"][gid].to(torch.float32)
    order = torch.argsort(d32, stable=True)
    order = order[torch.argsort(tile[order], stable=True)]
    tile, gid = tile[order], gid[order]
    starts = torch.searchsorted(tile, torch.arange(gx * gy))
    ends = torch.searchsorted(tile, torch.arange(gx * gy), right=True)

    feat, m2, co, dep = g["rgb"], g["means2D"], g["conic_opacity"], g["depths"]
    ly, lx = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
    for t in range(gx * gy):
        s, e = int(starts[t]), int(ends[t])
        ty, tx = divmod(t, gx)
        px = (tx * TILE + lx).reshape(-1)
        py = (ty * TILE + ly).reshape(-1)
        inside = (px < W) & (py < H)
        px, py = px[inside], py[inside]
        npx = px.numel()
        T = torch.ones(npx, dtype=dtype)
        C = torch.zeros(npx, 3, dtype=dtype)
        Dacc = torch.zeros(npx, dtype=dtype)
        alive = torch.ones(npx, dtype=torch.bool)
        for c0 in range(s, e, chunk):
            if not bool(alive.any()):
                break
            gsel = gid[c0:min(e, c0 + chunk)]
            dx = m2[gsel, 0][:, None] - px.to(dtype)[None]
            dy = m2[gsel, 1][:, None] - py.to(dtype)[None]
            cc = co[gsel]
            power = -0.5 * (cc[:, 0:1] * dx * dx + cc[:, 2:3] * dy * dy) - cc[:, 1:2] * dx * dy
            a = torch.clamp_max(cc[:, 3:4] * torch.exp(power), 0.99)
            a = torch.where((power > 0) | (a < 1.0 / 255.0), torch.zeros((), dtype=dtype), a)
            Tin = torch.cumprod(torch.cat((T[None], 1 - a), 0), 0)       # [L+1, npx]; row k = T before entry k
            sat = (a > 0) & (Tin[1:] < 0.0001)
            stop = torch.where(sat.any(0), sat.to(torch.int8).argmax(0), a.shape[0])  # first saturating entry
            use = (torch.arange(a.shape[0])[:, None] < stop[None]) & alive[None]
            w = torch.where(use, a * Tin[:-1], torch.zeros((), dtype=dtype))
            C += torch.einsum("lp,lc->pc", w, feat[gsel])
            Dacc += (w * dep[gsel][:, None]).sum(0)
            Tnew = Tin[stop.clamp(max=a.shape[0]), torch.arange(npx)]
            T = torch.where(alive, Tnew, T)
            alive = alive & (stop >= a.shape[0])
        color[:, py, px] = (C + T[:, None] * bgv[None]).t()
        depth[0, py, px] = Dacc
        alpha[0, py, px] = 1 - T
    return {"color": color, "depth": depth, "alpha": alpha, "radii": g["radii"], "num_rendered": D,
            "point_list": gid, "tile_ids": tile}
