"""Seeded synthetic Gaussian clouds for the BASELINE.md configs (C1-C4).

There is no network for real scenes or checkpoints, so every measurement and parity case uses
clouds drawn from the distributions fixed in BASELINE.md section 3 / SURVEY.md section 8d.  Tensors
are in the *activated* form the rasterizer receives from ``GaussianModel``'s getters
(``sugar/gaussian_splatting/scene/gaussian_model.py:95-128``): scales already ``exp``-ed,
rotations unit (w,x,y,z), opacities already ``sigmoid``-ed, SH as ``[P, M, 3]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from .cameras import Camera


@dataclass
class GaussianCloud:
    means3D: torch.Tensor            # [P,3]
    opacities: torch.Tensor          # [P,1]
    scales: torch.Tensor             # [P,3]
    rotations: torch.Tensor          # [P,4] (w,x,y,z), unit
    shs: Optional[torch.Tensor]      # [P,M,3] or None
    colors_precomp: Optional[torch.Tensor] = None  # [P,3] or None
    sh_degree: int = 3

    @property
    def P(self) -> int:
        return int(self.means3D.shape[0])

    def to(self, device) -> "GaussianCloud":
        mv = lambda t: None if t is None else t.to(device).contiguous()
        return GaussianCloud(mv(self.means3D), mv(self.opacities), mv(self.scales), mv(self.rotations),
                             mv(self.shs), mv(self.colors_precomp), self.sh_degree)

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in
                   (self.means3D, self.opacities, self.scales, self.rotations, self.shs, self.colors_precomp)
                   if t is not None)


def _cloud(P: int, seed: int, xyz_kind: str, log_scale_mu: float, log_scale_sigma: float,
           sh_coeffs: int = 16, sh_degree: int = 3) -> GaussianCloud:
    g = torch.Generator(device="cpu").manual_seed(seed)
    if xyz_kind == "uniform":
        xyz = torch.rand(P, 3, generator=g) * 2.0 - 1.0
    else:
        xyz = torch.randn(P, 3, generator=g).clamp_(-3.0, 3.0)
    scales = torch.exp(torch.randn(P, 3, generator=g) * log_scale_sigma + log_scale_mu)
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5)
    dc = torch.randn(P, 1, 3, generator=g)
    rest = torch.randn(P, sh_coeffs - 1, 3, generator=g) * 0.2
    shs = torch.cat((dc, rest), dim=1).contiguous()
    return GaussianCloud(xyz.contiguous(), opac.contiguous(), scales.contiguous(), q.contiguous(), shs,
                         None, sh_degree)


def config_c1(P: int = 10_000, seed: int = 0) -> GaussianCloud:
    """C1: uniform cube, sigma ~ 0.03."""
    return _cloud(P, seed, "uniform", math.log(0.03), 0.3)


def config_c2(P: int = 1_000_000, seed: int = 1) -> GaussianCloud:
    """C2 (Garden stand-in): clipped normal cloud, sigma ~ 0.008."""
    return _cloud(P, seed, "normal", math.log(0.008), 0.6)


def config_c3(P: int = 3_000_000, seed: int = 2) -> GaussianCloud:
    """C3 (headline): clipped normal cloud, sigma ~ 0.005."""
    return _cloud(P, seed, "normal", math.log(0.005), 0.6)


def config_heavy(P: int = 1_000_000, seed: int = 5, log_scale_mu: float = math.log(0.0045), log_scale_sigma: float = 1.2,
                 max_anisotropy: float = 10.0) -> GaussianCloud:
    """A trained-scene-like stress cloud (not a BASELINE config): what optimised scenes look like and the C2 / C3
    stand-ins do not -- heavy-tailed sizes (log-normal, sigma 1.2: a few splats cover a third of the screen), needle /
    disc anisotropy up to 10:1 (log-uniform stretch of one axis, the third axis squeezed), bimodal opacity (60 % near
    opaque, 40 % faint).  At 960x540 that is ~20 (tile, Gaussian) pairs of the reference per Gaussian and per-tile lists
    in the thousands, most of them from splats whose bounding square covers hundreds of tiles."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    xyz = torch.randn(P, 3, generator=g).clamp_(-3.0, 3.0)
    base = torch.randn(P, 1, generator=g) * log_scale_sigma + log_scale_mu
    stretch = torch.rand(P, 1, generator=g) * math.log(max_anisotropy)
    scales = torch.exp(torch.cat((base + stretch, base, base - 0.5 * stretch), dim=1))
    q = torch.randn(P, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    solid = torch.rand(P, 1, generator=g) < 0.6
    opac = torch.where(solid, torch.sigmoid(torch.randn(P, 1, generator=g) * 0.7 + 3.0),
                       torch.sigmoid(torch.randn(P, 1, generator=g) * 0.7 - 3.0))
    dc = torch.randn(P, 1, 3, generator=g)
    rest = torch.randn(P, 15, 3, generator=g) * 0.2
    return GaussianCloud(xyz.contiguous(), opac.contiguous(), scales.contiguous(), q.contiguous(),
                         torch.cat((dc, rest), dim=1).contiguous(), None, 3)


def config_c4(P: int = 200_000, seed: int = 3, extent: float = 3.0) -> GaussianCloud:
    """C4: SuGaR-style flat, surface-bound Gaussians with precomputed colours.

    Mirrors the call shape of ``sugar/sugar_scene/sugar_model.py:409-452,2141-2183``: the first
    scale axis is a thin ``surface_mesh_thickness`` (1e-4 * extent), the other two are in-plane,
    quaternions come from random tangent frames on a sphere of radius 1.5, colours are passed as
    ``colors_precomp`` (no SH).
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    n = torch.randn(P, 3, generator=g)
    n = n / n.norm(dim=1, keepdim=True)
    xyz = n * 1.5
    helper = torch.tensor([0.0, 0.0, 1.0]).expand(P, 3).clone()
    helper[n[:, 2].abs() > 0.9] = torch.tensor([1.0, 0.0, 0.0])
    t1 = torch.linalg.cross(n, helper)
    t1 = t1 / t1.norm(dim=1, keepdim=True)
    t2 = torch.linalg.cross(n, t1)
    R = torch.stack((n, t1, t2), dim=2)  # columns: normal, tangent, bitangent
    q = _rotmat_to_quat(R)
    s_in = torch.exp(torch.randn(P, 2, generator=g) * 0.4 + math.log(0.02))
    scales = torch.cat((torch.full((P, 1), 1e-4 * extent), s_in), dim=1)
    opac = torch.sigmoid(torch.randn(P, 1, generator=g) * 1.5 + 1.0)
    colors = torch.rand(P, 3, generator=g)
    return GaussianCloud(xyz.contiguous(), opac.contiguous(), scales.contiguous(), q.contiguous(), None,
                         colors.contiguous(), 0)


def _rotmat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """Batched rotation matrix -> unit quaternion (w,x,y,z); numerically safe branchless form."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    w = torch.sqrt(torch.clamp(1 + m00 + m11 + m22, min=0)) / 2
    x = torch.sqrt(torch.clamp(1 + m00 - m11 - m22, min=0)) / 2
    y = torch.sqrt(torch.clamp(1 - m00 + m11 - m22, min=0)) / 2
    z = torch.sqrt(torch.clamp(1 - m00 - m11 + m22, min=0)) / 2
    x = torch.copysign(x, R[:, 2, 1] - R[:, 1, 2])
    y = torch.copysign(y, R[:, 0, 2] - R[:, 2, 0])
    z = torch.copysign(z, R[:, 1, 0] - R[:, 0, 1])
    q = torch.stack((w, x, y, z), dim=1)
    return q / q.norm(dim=1, keepdim=True)


def c1_camera(width: int = 256, height: int = 256, fovx_deg: float = 60.0) -> Camera:
    """C1 camera: at (0,0,-4), identity rotation, looking down +z."""
    fovx = math.radians(fovx_deg)
    fovy = 2.0 * math.atan(math.tan(fovx / 2.0) * height / width)
    return Camera.from_Rt(np.eye(3), np.array([0.0, 0.0, 4.0]), fovx, fovy, width, height, "00000")
