"""Probe of the PyTorch-ROCm kernels the reference's per-frame ``render()`` prep runs (gaussian_model.py:95-128,
general_utils.py:78-157, gaussian_renderer/__init__.py:118-208): dumps inputs, intermediates and outputs of every
operation on the GPU, so that the rounding / summation order of each one can be identified OFFLINE
(scripts/experiments/torch_op_identify.py) and restated inside the fused raw-parameter kernels bit for bit.

Run on the GPU box: ``python scripts/experiments/torch_op_probe.py gpurun_out/torch_probe``.
"""
import sys

import numpy as np
import torch
import torch.nn.functional as F


def build_rotation(r):
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    r_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r_ * z)
    R[:, 0, 2] = 2 * (x * z + r_ * y)
    R[:, 1, 0] = 2 * (x * y + r_ * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r_ * x)
    R[:, 2, 0] = 2 * (x * z - r_ * y)
    R[:, 2, 1] = 2 * (y * z + r_ * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R, norm, q


def per_gaussian(P, seed, dev, rows):
    g = torch.Generator().manual_seed(seed)
    xyz = (torch.randn(P, 3, generator=g) * 2.0)
    campos = torch.tensor([0.3, -1.7, 4.1])
    ls = torch.randn(P, 3, generator=g) * 0.6 + float(np.log(0.005))
    rot = torch.randn(P, 4, generator=g)
    rot[1::7] *= 1e-3
    rot[3::11] *= 50.0
    op = torch.randn(P, 1, generator=g) * 1.5
    op[0::13] *= 10.0
    op[5::97] = 95.0
    op[6::97] = -95.0
    # scale ties: two equal minima / all three equal (argsort's tie order)
    ls[2::17, 1] = ls[2::17, 0]
    ls[4::19, 2] = ls[4::19, 0]
    ls[6::23, 1] = ls[6::23, 0]; ls[6::23, 2] = ls[6::23, 0]
    ls[8::29, 2] = ls[8::29, 1]
    xyz, campos, ls, rot, op = (t.to(dev) for t in (xyz, campos, ls, rot, op))
    out = {"xyz": xyz, "campos": campos, "log_scale": ls, "rot_raw": rot, "opacity_raw": op}
    out["scales"] = torch.exp(ls)
    out["opacity"] = torch.sigmoid(op)
    out["rot_norm"] = rot.norm(2, 1, keepdim=True)
    out["rot"] = F.normalize(rot)
    dir_pp = xyz - campos.repeat(P, 1)
    out["dir_pp"] = dir_pp
    out["dir_norm"] = dir_pp.norm(dim=1, keepdim=True)
    dirn = dir_pp / out["dir_norm"]
    out["dirn"] = dirn
    R, bn, bq = build_rotation(out["rot"])
    out["R"], out["build_norm"], out["build_q"] = R, bn, bq
    idx = torch.argsort(out["scales"], descending=False, dim=-1)
    out["argsort"] = idx.to(torch.int32)
    R_sorted = torch.gather(R, dim=2, index=idx[:, None, :].repeat(1, 3, 1)).squeeze()
    axis = R_sorted[:, :, 0]
    out["axis"] = axis.contiguous()
    dot = torch.sum(axis * -dirn, dim=-1, keepdims=True)
    out["dot"] = dot
    non_flip = dot >= 0
    flipped = axis * torch.where(non_flip, 1, -1)
    out["flipped"] = flipped
    out["flip_norm"] = flipped.norm(dim=1, keepdim=True)
    normal = flipped / out["flip_norm"]
    out["normal"] = normal
    out["normal_normed"] = normal * 0.5 + 0.5
    if str(dev) != "cpu":
        torch.cuda.synchronize()
    res = {}
    for k, v in out.items():
        v = v.detach().cpu().numpy()
        res[k] = v[rows] if (rows is not None and v.ndim >= 1 and v.shape[0] == P) else v
    if rows is not None:
        res["rows"] = rows
    return res


def per_pixel(H, W, seed, dev, crop):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(3, H, W, generator=g)
    img[:, ::5, ::7] = 0.5          # zero vectors after (x - 0.5) * 2: F.normalize's eps path
    depth = torch.rand(H, W, generator=g) * 6.0
    depth[::9, ::4] = 0.0
    fovx = 60.0 * np.pi / 180.0
    fx = W / (2 * np.tan(fovx / 2))
    fovy = 2 * np.arctan(H / (2 * fx))
    fy = H / (2 * np.tan(fovy / 2))
    cx, cy = W / 2, H / 2
    A = torch.randn(3, 3, generator=g)
    Q, _ = torch.linalg.qr(A)
    w2c = torch.eye(4)
    w2c[:3, :3] = Q
    w2c[:3, 3] = torch.tensor([0.2, -0.4, 3.7])
    wvt = w2c.T.contiguous()
    img, depth, wvt = img.to(dev), depth.to(dev), wvt.to(dev)
    out = {"img": img, "depth": depth, "world_view_transform": wvt,
           "intr": torch.tensor([fx, fy, cx, cy], dtype=torch.float64)}
    t = (img - 0.5) * 2.
    out["img_pm1"] = t
    out["normal_image"] = F.normalize(t.permute(1, 2, 0), p=2, dim=-1)
    c2w = wvt.inverse()
    out["c2w"] = c2w
    K = torch.FloatTensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]])
    v, u = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev), indexing="ij")
    fx_, fy_, cx_, cy_ = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    directions = torch.stack([(u - cx_ + 0.5) / fx_, (v - cy_ + 0.5) / fy_, torch.ones_like(u)], -1)
    out["directions"] = directions
    directions_py = torch.stack(((u - cx + 0.5) / fx, (v - cy + 0.5) / fy, torch.ones_like(u)), -1)
    out["directions_py"] = directions_py
    rays_d = directions @ c2w[:3, :3].T
    out["rays_d"] = rays_d
    rays_o = c2w[:3, 3].expand_as(rays_d)
    points3D = rays_o + rays_d * depth.unsqueeze(-1)
    out["points3D"] = points3D
    hd, wd, _ = points3D.shape
    bottom, top = points3D[2:hd, 1:wd - 1, :], points3D[0:hd - 2, 1:wd - 1, :]
    right, left = points3D[1:hd - 1, 2:wd, :], points3D[1:hd - 1, 0:wd - 2, :]
    l2r, b2t = right - left, top - bottom
    cr = torch.cross(l2r, b2t, dim=-1)
    out["cross"] = cr
    n = F.normalize(cr, p=2, dim=-1)
    out["cross_unit"] = n
    out["pseudo_normal"] = F.pad(n.permute(2, 0, 1), (1, 1, 1, 1), mode="constant").permute(1, 2, 0)
    if str(dev) != "cpu":
        torch.cuda.synchronize()
    res = {}
    for k, v in out.items():
        v = v.detach().cpu().numpy()
        if crop is not None and v.ndim >= 2:
            r0, r1 = crop
            if v.shape[:2] == (H, W):
                v = v[r0:r1]
            elif v.shape[:2] == (H - 2, W - 2):
                v = v[r0:r1 - 2]        # row i of the inner arrays is image row i + 1
            elif v.ndim == 3 and v.shape[1:] == (H, W):
                v = v[:, r0:r1]
        res[k] = np.ascontiguousarray(v)
    if crop is not None:
        res["crop"] = np.array(crop)
    res["HW"] = np.array([H, W])
    return res


def main():
    outdir = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/torch_probe"
    import os
    os.makedirs(outdir, exist_ok=True)
    dev = os.environ.get("PROBE_DEVICE", "cuda:0")
    print("torch", torch.__version__, torch.cuda.get_device_name(0) if dev != "cpu" else "cpu")
    np.savez(os.path.join(outdir, "pg_small.npz"), **per_gaussian(40_003, 1, dev, None))
    P = 3_000_000
    rows = np.concatenate([np.arange(0, 3000), np.random.default_rng(0).integers(0, P, 3000), np.arange(P - 3000, P)])
    np.savez(os.path.join(outdir, "pg_big.npz"), **per_gaussian(P, 2, dev, rows))
    np.savez(os.path.join(outdir, "px_small.npz"), **per_pixel(120, 200, 3, dev, None))
    np.savez(os.path.join(outdir, "px_odd.npz"), **per_pixel(61, 97, 5, dev, None))
    np.savez(os.path.join(outdir, "px_big.npz"), **per_pixel(1080, 1920, 4, dev, (500, 540)))
    print("done")


if __name__ == "__main__":
    main()
