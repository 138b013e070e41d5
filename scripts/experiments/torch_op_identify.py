"""Offline half of torch_op_probe.py: which fp32 formula (summation order, fused or not, division or reciprocal) each
PyTorch-ROCm kernel of the reference's per-frame prep evaluates.  Every candidate is computed in numpy with one rounding
per operation (fma emulated through float64: exact product, one rounding of the sum, then to float32) and compared bit
for bit with what the GPU produced.  ``python scripts/experiments/torch_op_identify.py gpurun_out/torch_probe``.
"""
import sys

import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def rate(name, got, want):
    got, want = np.asarray(got), np.asarray(want)
    same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
    print(f"    {name:58s} {same.mean() * 100:9.5f} %   mismatches {int((~same).sum())} / {same.size}")
    return same


def norm_candidates(x):
    """sqrt of the sum of squares of the last axis (3 or 4 long) in every plausible order."""
    sq = [x[..., k] * x[..., k] for k in range(x.shape[-1])]
    out = {}
    if len(sq) == 3:
        out["(0+1)+2"] = np.sqrt((sq[0] + sq[1]) + sq[2])
        out["(0+2)+1"] = np.sqrt((sq[0] + sq[2]) + sq[1])
        out["0+(1+2)"] = np.sqrt(sq[0] + (sq[1] + sq[2]))
        out["fma chain 0,1,2"] = np.sqrt(fma(x[..., 2], x[..., 2], fma(x[..., 1], x[..., 1], sq[0])))
        out["fma (0,2),1"] = np.sqrt(fma(x[..., 2], x[..., 2], sq[0]) + sq[1])
    else:
        out["((0+1)+2)+3"] = np.sqrt(((sq[0] + sq[1]) + sq[2]) + sq[3])
        out["(0+1)+(2+3)"] = np.sqrt((sq[0] + sq[1]) + (sq[2] + sq[3]))
        out["(0+2)+(1+3)"] = np.sqrt((sq[0] + sq[2]) + (sq[1] + sq[3]))
        out["fma chain"] = np.sqrt(fma(x[..., 3], x[..., 3], fma(x[..., 2], x[..., 2], fma(x[..., 1], x[..., 1], sq[0]))))
    return out


def per_gaussian(d):
    print("  exp / sigmoid (numpy's exp is not the device's: informational)")
    rate("scales == np.exp", d["scales"], np.exp(d["log_scale"]))
    e = np.exp(-d["opacity_raw"].astype(f32))
    rate("opacity == 1/(1+np.exp(-x))", d["opacity"], f32(1) / (f32(1) + e))
    print("  rot.norm(2, dim=1) [P,4]")
    for k, v in norm_candidates(d["rot_raw"]).items():
        rate(k, d["rot_norm"][:, 0], v)
    print("  F.normalize(rot) given the GPU's norm")
    den = np.maximum(d["rot_norm"], f32(1e-12))
    rate("x / max(norm, 1e-12)", d["rot"], d["rot_raw"] / den)
    rate("x * (1 / max(norm, 1e-12))", d["rot"], d["rot_raw"] * (f32(1) / den))
    print("  dir_pp = xyz - campos")
    rate("xyz - campos", d["dir_pp"], d["xyz"] - d["campos"][None])
    print("  dir_pp.norm(dim=1) [P,3]")
    for k, v in norm_candidates(d["dir_pp"]).items():
        rate(k, d["dir_norm"][:, 0], v)
    rate("dirn = dir_pp / norm", d["dirn"], d["dir_pp"] / d["dir_norm"])
    print("  build_rotation(normalised rot)")
    r = d["rot"]
    bn = np.sqrt(((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2]) + r[:, 3] * r[:, 3])
    rate("norm sequential", d["build_norm"], bn)
    q = r / d["build_norm"][:, None]
    rate("q = r / norm", d["build_q"], q)
    w, x, y, z = (d["build_q"][:, k] for k in range(4))
    one, two = f32(1), f32(2)
    R = np.zeros((r.shape[0], 3, 3), f32)
    R[:, 0, 0] = one - two * (y * y + z * z); R[:, 0, 1] = two * (x * y - w * z); R[:, 0, 2] = two * (x * z + w * y)
    R[:, 1, 0] = two * (x * y + w * z); R[:, 1, 1] = one - two * (x * x + z * z); R[:, 1, 2] = two * (y * z - w * x)
    R[:, 2, 0] = two * (x * z - w * y); R[:, 2, 1] = two * (y * z + w * x); R[:, 2, 2] = one - two * (x * x + y * y)
    rate("R unfused", d["R"], R)
    print("  argsort(scales) tie order / minimum axis")
    s = d["scales"]
    first = np.where((s[:, 0] <= s[:, 1]) & (s[:, 0] <= s[:, 2]), 0, np.where(s[:, 1] <= s[:, 2], 1, 2))
    last = np.where((s[:, 2] <= s[:, 1]) & (s[:, 2] <= s[:, 0]), 2, np.where(s[:, 1] <= s[:, 0], 1, 0))
    ties = (s[:, 0] == s[:, 1]) | (s[:, 0] == s[:, 2]) | (s[:, 1] == s[:, 2])
    mins = d["argsort"][:, 0]
    print(f"    rows with tied scales: {int(ties.sum())}; argsort[:,0] == first minimum: {(mins == first).mean() * 100:.4f} %  "
          f"(tied rows: {(mins[ties] == first[ties]).mean() * 100:.4f} %); == last minimum: {(mins == last).mean() * 100:.4f} %")
    stable = np.argsort(s, axis=-1, kind="stable")
    print(f"    full argsort == stable argsort: {(d['argsort'] == stable).all(axis=1).mean() * 100:.4f} %")
    ax = np.take_along_axis(d["R"], mins[:, None, None].astype(np.int64).repeat(3, 1), axis=2)[:, :, 0]
    rate("axis == R[:, :, argmin]", d["axis"], ax)
    print("  dot = sum(axis * -dirn)")
    p = d["axis"] * -d["dirn"]
    rate("(0+1)+2", d["dot"][:, 0], (p[:, 0] + p[:, 1]) + p[:, 2])
    rate("(0+2)+1", d["dot"][:, 0], (p[:, 0] + p[:, 2]) + p[:, 1])
    rate("0+(1+2)", d["dot"][:, 0], p[:, 0] + (p[:, 1] + p[:, 2]))
    sgn = np.where(d["dot"] >= 0, f32(1), f32(-1))
    rate("flipped", d["flipped"], d["axis"] * sgn)
    print("  flipped.norm(dim=1)")
    for k, v in norm_candidates(d["flipped"]).items():
        rate(k, d["flip_norm"][:, 0], v)
    rate("normal = flipped / norm", d["normal"], d["flipped"] / d["flip_norm"])
    rate("normal * 0.5 + 0.5 unfused", d["normal_normed"], d["normal"] * f32(0.5) + f32(0.5))
    rate("fma(normal, 0.5, 0.5)", d["normal_normed"], fma(d["normal"], np.full_like(d["normal"], 0.5), np.full_like(d["normal"], 0.5)))


def per_pixel(d):
    H, W = (int(v) for v in d["HW"])
    crop = tuple(int(v) for v in d["crop"]) if "crop" in d else (0, H)
    r0, r1 = crop
    print("  (img - 0.5) * 2")
    t = (d["img"] - f32(0.5)) * f32(2)
    rate("unfused", d["img_pm1"], t)
    print("  F.normalize(permuted [H,W,3], dim=-1)")
    tt = np.moveaxis(d["img_pm1"], 0, -1)
    for k, v in norm_candidates(tt).items():
        rate(k, d["normal_image"], tt / np.maximum(v, f32(1e-12))[..., None])
    print("  directions")
    fx, fy, cx, cy = d["intr"]
    v, u = np.meshgrid(np.arange(r0, r1, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
    for nm, key in (("K tensor", "directions"), ("python floats", "directions_py")):
        g = d[key]
        rate(f"{nm}: x (u - cx + 0.5) * (1/fx)", g[..., 0], (u - f32(cx) + f32(0.5)) * (f32(1) / f32(fx)))
        rate(f"{nm}: x (u - cx + 0.5) / fx", g[..., 0], (u - f32(cx) + f32(0.5)) / f32(fx))
        rate(f"{nm}: y (v - cy + 0.5) * (1/fy)", g[..., 1], (v - f32(cy) + f32(0.5)) * (f32(1) / f32(fy)))
        rate(f"{nm}: y (v - cy + 0.5) / fy", g[..., 1], (v - f32(cy) + f32(0.5)) / f32(fy))
    print("  rays_d = directions @ c2w[:3,:3].T")
    D, M = d["directions"], d["c2w"][:3, :3]
    for j in range(3):
        a0, a1, a2 = D[..., 0], D[..., 1], D[..., 2]
        m0, m1, m2 = (np.full_like(a0, M[j, k]) for k in range(3))
        rate(f"col {j}: (a0 m0 + a1 m1) + a2 m2", d["rays_d"][..., j], (a0 * m0 + a1 * m1) + a2 * m2)
        rate(f"col {j}: fma(a2,m2, fma(a1,m1, a0 m0))", d["rays_d"][..., j], fma(a2, m2, fma(a1, m1, a0 * m0)))
        rate(f"col {j}: fma(a0,m0, fma(a1,m1, a2 m2))", d["rays_d"][..., j], fma(a0, m0, fma(a1, m1, a2 * m2)))
        rate(f"col {j}: fma(a1,m1, a0 m0) + a2 m2", d["rays_d"][..., j], fma(a1, m1, a0 * m0) + a2 * m2)
    print("  points3D = rays_o + rays_d * depth")
    ro = d["c2w"][:3, 3]
    rate("unfused", d["points3D"], ro[None, None] + d["rays_d"] * d["depth"][..., None])
    rate("fma", d["points3D"], fma(d["rays_d"], np.broadcast_to(d["depth"][..., None], d["rays_d"].shape).copy(),
                                   np.broadcast_to(ro[None, None], d["rays_d"].shape).copy()))
    print("  cross(right - left, top - bottom)")
    P3 = d["points3D"]
    hd, wd = P3.shape[:2]
    bottom, top = P3[2:hd, 1:wd - 1], P3[0:hd - 2, 1:wd - 1]
    right, left = P3[1:hd - 1, 2:wd], P3[1:hd - 1, 0:wd - 2]
    a, b = right - left, top - bottom
    idx = [(1, 2), (2, 0), (0, 1)]
    got = d["cross"]
    for name, fn in (("a_i b_j - a_j b_i unfused", lambda p, q, r, s: p * q - r * s),
                     ("fma(a_i, b_j, -(a_j b_i))", lambda p, q, r, s: fma(p, q, -(r * s))),
                     ("fma(-a_j, b_i, a_i b_j)", lambda p, q, r, s: fma(-r, s, p * q))):
        c = np.stack([fn(a[..., i], b[..., j], a[..., j], b[..., i]) for i, j in idx], -1)
        rate(name, got, c)
    print("  F.normalize(cross, dim=-1) contiguous")
    for k, v in norm_candidates(got).items():
        rate(k, d["cross_unit"], got / np.maximum(v, f32(1e-12))[..., None])


def main():
    root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/torch_probe"
    for name in ("pg_small", "pg_big"):
        print(name)
        per_gaussian(dict(np.load(f"{root}/{name}.npz")))
    for name in ("px_small", "px_odd", "px_big"):
        print(name)
        per_pixel(dict(np.load(f"{root}/{name}.npz")))


if __name__ == "__main__":
    main()
