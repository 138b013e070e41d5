"""TEST / BENCH INFRASTRUCTURE -- the reference's per-frame scene composition carried out with PyTorch tensor operations on
whatever device the models live on, in the reference's own sequence (``/root/reference/gaussians_utils.py:71-118``:
``transform_gaussians`` then ``merge_two_gaussians``; ``/root/reference/rotation_utils.py:113-150`` for the quaternion
product; ``sugar/gaussian_splatting/scene/gaussian_model.py:95-128`` for the activations ``render()`` applies afterwards),
minus the deep copy of Python objects and the per-frame PLY reload.

It is the measuring stick ``bench.py`` times beside ``autovfx_amd.dynamic_scene.DynamicScene`` (``also.c5_dynamic``:
"the reference's structure in PyTorch on the same GPU") and the on-GPU parity partner of tests/test_dynamic_scene.py.  Like
everything under ``oracle/`` it is never imported by the product.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from autovfx_amd.dynamic_scene import matrix_to_quaternion
from autovfx_amd.scenes import GaussianCloud


def reference_shaped_compose(base, objects: Dict[str, Tuple[object, Sequence[float]]], placements, device,
                             reference_sh_degree: bool = True) -> GaussianCloud:
    """The same frame composed the reference's way with PyTorch on the GPU -- clone the base parameters, transform each
    placed object's raw parameters with the reference's sequence of tensor operations (``gaussians_utils.py:85-118``),
    concatenate everything (``:71-82``), activate (``gaussian_model.py:95-128``) -- minus the per-frame PLY reload.  The
    measuring stick for ``DynamicScene`` (bench.py ``also.c5_dynamic``) and its on-GPU parity partner (tests)."""
    t = lambda a: a.detach().to(device=device, dtype=torch.float32)
    parts = {k: [t(getattr(base, k)).clone()] for k in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest")}
    for entry in placements:
        obj_id, center, rotation, scaling = entry[:4]
        m, c0 = objects[obj_id]
        if len(entry) > 4 and entry[4] is not None:   # the melting branch (scene_representation.py:409-417): boolean-mask indexing
            mask = torch.as_tensor(np.asarray(entry[4]), device=device)
            sub = {k: t(getattr(m, k))[mask] for k in ("_xyz", "_rotation", "_scaling", "_opacity", "_features_dc", "_features_rest")}
            if center is None:   # merged as it is
                for k, v in sub.items():
                    parts[k].append(v)
                continue
            m = type("Subset", (), sub)
        c0 = torch.as_tensor(np.asarray(c0, np.float32), device=device)
        center = torch.as_tensor(np.asarray(center, np.float32), device=device)
        R = torch.as_tensor(np.asarray(rotation, np.float32).reshape(3, 3), device=device)
        xyz, rot, ls = t(m._xyz).clone(), t(m._rotation).clone(), t(m._scaling).clone()
        xyz -= c0.unsqueeze(0); xyz *= scaling; xyz += c0.unsqueeze(0)
        ls += np.log(scaling)
        xyz -= c0.unsqueeze(0)
        xyz = torch.matmul(xyz, R.T)
        xyz += c0.unsqueeze(0)
        qR = torch.as_tensor(matrix_to_quaternion(R.cpu().numpy()), device=device)
        aw, ax, ay, az = torch.unbind(qR.expand_as(rot), -1)
        bw, bx, by, bz = torch.unbind(rot, -1)
        q = torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                         aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)
        q = torch.where(q[..., 0:1] < 0, -q, q)
        xyz += (center - c0).unsqueeze(0)
        for k, v in (("_xyz", xyz), ("_rotation", q), ("_scaling", ls), ("_opacity", t(m._opacity)), ("_features_dc", t(m._features_dc)),
                     ("_features_rest", t(m._features_rest))):
            parts[k].append(v)
    cat = {k: torch.cat(v, dim=0) for k, v in parts.items()}
    return GaussianCloud(cat["_xyz"].contiguous(), torch.sigmoid(cat["_opacity"]).contiguous(), torch.exp(cat["_scaling"]).contiguous(),
                         torch.nn.functional.normalize(cat["_rotation"]).contiguous(),
                         torch.cat((cat["_features_dc"], cat["_features_rest"]), dim=1).contiguous(), None,
                         # a merged model is a fresh GaussianModel: active_sh_degree 0 (gaussians_utils.py:75, gaussian_model.py:49)
                         (0 if reference_sh_degree and len(placements) else int(getattr(base, "active_sh_degree", 3))))
