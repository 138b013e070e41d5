"""TEST INFRASTRUCTURE -- CPU restatement (numpy, fp32, every operation a separate rounding) of the per-frame scene
composition the reference performs before it renders an edited scene:

* ``transform_gaussians``      /root/reference/gaussians_utils.py:85-118   (scale, rotate, translate about ``initial_center``)
* ``merge_two_gaussians``      /root/reference/gaussians_utils.py:71-82    (concatenation, base first)
* ``matrix_to_quaternion``, ``quaternion_multiply``, ``standardize_quaternion``
                               /root/reference/rotation_utils.py:24-85, 113-135, 137-150
* the activations ``render()`` applies to the merged model
                               /root/reference/sugar/gaussian_splatting/scene/gaussian_model.py:95-128

Only tests/, ``__graft_entry__.smoke()`` and bench.py's baseline legs may import this module; the product
(``autovfx_amd.dynamic_scene`` -> ``gsr_place_object``) never does.  Pinned against the reference's own functions executed in
PyTorch on the CPU by tests/test_dynamic_scene.py when /root/reference is mounted: the quaternion path and the log-scales
bit for bit; positions to 2 ulp (``torch.matmul``'s summation order inside a BLAS call is not specified -- this restatement
and the HIP kernel fix it as ``(x0 r0 + x1 r1) + x2 r2``); ``exp`` to 1 ulp (libm implementations differ in the last bit).
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def matrix_to_quaternion(R) -> np.ndarray:
    """rotation_utils.py:24-85 for one 3x3 matrix: (w, x, y, z), the best-conditioned of the four candidates."""
    m = np.asarray(R, dtype=f32).reshape(9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = (f32(v) for v in m)
    one = f32(1.0)
    pre = np.array([one + m00 + m11 + m22, one + m00 - m11 - m22, one - m00 + m11 - m22, one - m00 - m11 + m22], dtype=f32)
    q_abs = np.where(pre > 0, np.sqrt(np.maximum(pre, f32(0))), f32(0)).astype(f32)
    sq = (q_abs * q_abs).astype(f32)   # q_abs ** 2
    cand = np.array([[sq[0], m21 - m12, m02 - m20, m10 - m01],
                     [m21 - m12, sq[1], m10 + m01, m02 + m20],
                     [m02 - m20, m10 + m01, sq[2], m12 + m21],
                     [m10 - m01, m20 + m02, m21 + m12, sq[3]]], dtype=f32)
    den = (f32(2.0) * np.maximum(q_abs, f32(0.1))).astype(f32)
    cand = (cand / den[:, None]).astype(f32)
    return cand[int(np.argmax(q_abs))]


def quaternion_multiply(a, b) -> np.ndarray:
    """rotation_utils.py:113-135: Hamilton product a (x) b of (w, x, y, z) rows, left-to-right sums, then the real part made
    non-negative (standardize_quaternion :137-150)."""
    a, b = np.asarray(a, dtype=f32), np.asarray(b, dtype=f32)
    aw, ax, ay, az = (a[..., k] for k in range(4))
    bw, bx, by, bz = (b[..., k] for k in range(4))
    ow = aw * bw - ax * bx - ay * by - az * bz
    ox = aw * bx + ax * bw + ay * bz - az * by
    oy = aw * by - ax * bz + ay * bw + az * bx
    oz = aw * bz + ax * by - ay * bx + az * bw
    out = np.stack((ow, ox, oy, oz), -1).astype(f32)
    return np.where(out[..., 0:1] < 0, -out, out).astype(f32)


def transform_raw(xyz, rotation, log_scale, center, R, scaling, initial_center):
    """transform_gaussians on the RAW parameters (gaussians_utils.py:85-118): returns (xyz, rotation, log_scale)."""
    xyz, rotation, log_scale = (np.asarray(v, dtype=f32) for v in (xyz, rotation, log_scale))
    c, c0, R = np.asarray(center, dtype=f32), np.asarray(initial_center, dtype=f32), np.asarray(R, dtype=f32).reshape(3, 3)
    s = f32(scaling)
    x = xyz - c0[None]
    x = x * s
    x = x + c0[None]
    ls = log_scale + f32(np.log(np.float64(scaling)))
    x = x - c0[None]
    x = np.stack([(x[:, 0] * R[j, 0] + x[:, 1] * R[j, 1]) + x[:, 2] * R[j, 2] for j in range(3)], axis=1).astype(f32)   # x @ R.T
    x = x + c0[None]
    q = quaternion_multiply(matrix_to_quaternion(R)[None], rotation)
    x = x + (c - c0)[None]
    return x.astype(f32), q, ls.astype(f32)


def activate(rotation, log_scale):
    """gaussian_model.py:96-101: scales = exp(raw), rotations = F.normalize(raw) (p = 2, eps = 1e-12)."""
    q = np.asarray(rotation, dtype=f32)
    # the sum of the four squares in the order the framework's reduction kernel takes them on the GPU the reference runs on
    # (four lanes, combined by shuffles: pairwise; identified on the device, scripts/experiments/torch_op_identify.py)
    norm = np.sqrt((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + (q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])).astype(f32)
    return (q / np.maximum(norm, f32(1e-12))[:, None]).astype(f32), np.exp(np.asarray(log_scale, dtype=f32)).astype(f32)


def placement_block(center, R, scaling, initial_center) -> np.ndarray:
    """The 21 floats gsr_place_object takes: c[3], R[9] row-major, s, c0[3], q_R[4], log_s."""
    R = np.asarray(R, dtype=f32).reshape(3, 3)
    return np.concatenate((np.asarray(center, f32), R.reshape(9), [f32(scaling)], np.asarray(initial_center, f32),
                           matrix_to_quaternion(R), [f32(np.log(np.float64(scaling)))])).astype(f32)


def compose(base, placements, base_sh_degree=3):
    """``base`` and every placed object: dicts of RAW arrays ``xyz, rotation, log_scale, opacity_raw, features_dc,
    features_rest``; ``placements``: list of ``(object, center, R, scaling, initial_center[, mask])`` in merge order; a mask selects a subset of
    the object, ``center = None`` merges it untransformed (the melting branch).  Returns the
    ACTIVATED arrays the rasterizer receives from the merged model: means3D, scales, rotations, opacities [P,1], shs [P,M,3]."""
    parts = {k: [np.asarray(base[k], dtype=f32)] for k in ("xyz", "rotation", "log_scale", "opacity_raw", "features_dc", "features_rest")}
    for entry in placements:
        obj, center, R, scaling, c0 = entry[:5]
        if len(entry) > 5 and entry[5] is not None:   # a subset of the object (scene_representation.py:409-416: `orig._xyz[mask]` ...)
            mask = np.asarray(entry[5])
            obj = {k: np.asarray(v)[mask] for k, v in obj.items()}
        if center is None:   # the melting branch merges the subset as it is: no transform_gaussians call (:417)
            x, q, ls = obj["xyz"], obj["rotation"], obj["log_scale"]
        else:
            x, q, ls = transform_raw(obj["xyz"], obj["rotation"], obj["log_scale"], center, R, scaling, c0)
        for k, v in (("xyz", x), ("rotation", q), ("log_scale", ls), ("opacity_raw", obj["opacity_raw"]),
                     ("features_dc", obj["features_dc"]), ("features_rest", obj["features_rest"])):
            parts[k].append(np.asarray(v, dtype=f32))
    cat = {k: np.concatenate(v, axis=0) for k, v in parts.items()}
    rot, scales = activate(cat["rotation"], cat["log_scale"])
    opac = (f32(1.0) / (f32(1.0) + np.exp(-cat["opacity_raw"]))).astype(f32)
    # The degree the reference renders this frame at: merge_two_gaussians builds a fresh GaussianModel (gaussians_utils.py:75)
    # whose active_sh_degree stays the constructor's 0 (scene/gaussian_model.py:49), and render() passes pc.active_sh_degree
    # (gaussian_renderer/__init__.py:111); without a placement the deep-copied scene keeps `base_sh_degree`.
    return {"means3D": cat["xyz"], "scales": scales, "rotations": rot, "opacities": opac.reshape(-1, 1),
            "shs": np.concatenate((cat["features_dc"], cat["features_rest"]), axis=1),
            "active_sh_degree": 0 if len(placements) else int(base_sh_degree)}
