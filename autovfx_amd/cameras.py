"""Camera matrices and trajectories in the conventions the rasterizer expects.

Host-side mirror of the reference's camera plumbing, so callers (tests, bench, the frame-parallel
driver) build `GaussianRasterizationSettings` from exactly the matrices AutoVFX would pass:

* world-to-view / projection / full-projection / camera-centre construction follows
  ``sugar/sugar_scene/cameras.py:212-221`` (``GSCamera``) and
  ``sugar/gaussian_splatting/utils/graphics_utils.py:39-78`` (``getWorld2View2``,
  ``getProjectionMatrix``, ``fov2focal``, ``focal2fov``);
* trajectory JSON and the half-sphere orbit follow ``dataset_utils/sample_custom_traj.py:44-106``
  and its consumer ``scene_representation.py:123-156``.

All matrices are stored *transposed* (row-major storage of the column-major matrix), which is
what the kernels index as ``m[4*col + row]``.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2.0 * math.tan(fov / 2.0))


def focal2fov(focal: float, pixels: float) -> float:
    return 2.0 * math.atan(pixels / (2.0 * focal))


def world_to_view(R: np.ndarray, t: np.ndarray, translate=(0.0, 0.0, 0.0), scale: float = 1.0) -> np.ndarray:
    """4x4 w2c as float32.  ``R`` is the *transposed* rotation (c2w rotation), as the reference stores it."""
    Rt = np.zeros((4, 4), dtype=np.float64)
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).T
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    c2w = np.linalg.inv(Rt)
    c2w[:3, 3] = (c2w[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.linalg.inv(c2w).astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """OpenGL-style perspective with z in [0,1] and w = +z (float32, un-transposed)."""
    top = math.tan(fovy / 2.0) * znear
    right = math.tan(fovx / 2.0) * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4, dtype=torch.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


@dataclass
class Camera:
    """The subset of ``GSCamera`` the render path reads."""

    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor  # [4,4] transposed w2c
    projection_matrix: torch.Tensor     # [4,4] transposed
    full_proj_transform: torch.Tensor   # [4,4] transposed (P @ w2c)^T
    camera_center: torch.Tensor         # [3]
    image_name: str = ""
    znear: float = 0.01
    zfar: float = 100.0
    # ``world_view_transform.inverse()``, which render() needs every frame for the pseudo normals
    # (gaussian_renderer/__init__.py:199); computing it on the GPU costs a host synchronisation per frame
    # (torch.linalg.inv reads its status word back), so it is formed once where the camera is built.
    view_world_transform: Optional[torch.Tensor] = None

    @property
    def tanfovx(self) -> float:
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.FoVy * 0.5)

    def to(self, device) -> "Camera":
        return Camera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.projection_matrix.to(device),
                      self.full_proj_transform.to(device), self.camera_center.to(device),
                      self.image_name, self.znear, self.zfar,
                      None if self.view_world_transform is None else self.view_world_transform.to(device))

    @staticmethod
    def batch_to(cams, device) -> list:
        """``[c.to(device) for c in cams]`` with FIVE host-to-device copies for the whole list instead of five per camera (a
        trajectory loop that uploads its camera inside the loop spends 0.6 ms per frame on these tiny copies): the matrices are
        stacked, copied once and handed out as views."""
        cams = list(cams)
        if not cams:
            return []
        stack = lambda ts: torch.stack([t.contiguous() for t in ts]).to(device)
        wv, pj = stack([c.world_view_transform for c in cams]), stack([c.projection_matrix for c in cams])
        full, center = stack([c.full_proj_transform for c in cams]), stack([c.camera_center for c in cams])
        have_inv = all(c.view_world_transform is not None for c in cams)
        inv = stack([c.view_world_transform for c in cams]) if have_inv else None
        return [Camera(c.image_width, c.image_height, c.FoVx, c.FoVy, wv[i], pj[i], full[i], center[i], c.image_name, c.znear, c.zfar,
                       inv[i] if have_inv else (None if c.view_world_transform is None else c.view_world_transform.to(device)))
                for i, c in enumerate(cams)]

    @staticmethod
    def from_Rt(R: np.ndarray, T: np.ndarray, FoVx: float, FoVy: float, width: int, height: int,
                name: str = "", znear: float = 0.01, zfar: float = 100.0) -> "Camera":
        wv = torch.tensor(world_to_view(R, T)).transpose(0, 1).contiguous()
        pj = projection_matrix(znear, zfar, FoVx, FoVy).transpose(0, 1).contiguous()
        full = wv.unsqueeze(0).bmm(pj.unsqueeze(0)).squeeze(0).contiguous()
        inv = wv.inverse().contiguous()
        center = inv[3, :3].contiguous()
        return Camera(int(width), int(height), float(FoVx), float(FoVy), wv, pj, full, center, name, znear, zfar, inv)

    @staticmethod
    def from_c2w(c2w: np.ndarray, fx: float, fy: float, width: int, height: int, name: str = "") -> "Camera":
        """The path ``scene_representation.py:141-156`` takes for a custom-trajectory frame."""
        w2c = np.linalg.inv(np.asarray(c2w, dtype=np.float64))
        R = np.transpose(w2c[:3, :3])
        T = w2c[:3, 3]
        return Camera.from_Rt(R, T, focal2fov(fx, width), focal2fov(fy, height), width, height, name)


def sugar_camera(c2w_opengl: np.ndarray, fov_x: float, fov_y: float, width: int, height: int, cx_ndc: float = 0.0,
                 cy_ndc: float = 0.0, name: str = "", znear: float = 0.01, zfar: float = 100.0) -> Camera:
    """The matrices SuGaR's ``render_image_gaussian_rasterizer`` hands to the rasterizer
    (``sugar/sugar_scene/sugar_model.py:2008-2032``): a nerfstudio camera-to-world matrix (OpenGL axes: y up, z back) is
    flipped to COLMAP axes (``c2w[:3, 1:3] *= -1``), inverted, and -- unlike the vanilla 3DGS camera -- the projection
    matrix carries the principal point of the pytorch3d camera: ``proj_transform[2, 0] = -K[0, 2]``,
    ``proj_transform[2, 1] = -K[1, 2]`` on the TRANSPOSED matrix (``cx_ndc``, ``cy_ndc``: pytorch3d NDC units, zero for a
    centred principal point).  ``viewmatrix`` and ``tanfov`` are unchanged, so the 2D covariance and the frustum clamp see the
    centred camera while the pixel centres are shifted: the call shape of BASELINE configs[3]."""
    c2w = np.array(c2w_opengl, dtype=np.float32)
    if c2w.shape == (3, 4):
        c2w = np.concatenate((c2w, np.array([[0, 0, 0, 1]], dtype=np.float32)), axis=0)
    c2w = c2w.copy()
    c2w[:3, 1:3] *= -1
    w2c = np.linalg.inv(c2w)
    R = np.transpose(w2c[:3, :3])
    T = w2c[:3, 3]
    Rt = np.zeros((4, 4), dtype=np.float32)      # getWorld2View (graphics_utils.py:39-50 without the translate / scale detour)
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = T
    Rt[3, 3] = 1.0
    wv = torch.tensor(Rt).transpose(0, 1).contiguous()
    pj = projection_matrix(znear, zfar, fov_x, fov_y).transpose(0, 1).contiguous()
    pj[2, 0] = -float(cx_ndc)
    pj[2, 1] = -float(cy_ndc)
    full = wv.unsqueeze(0).bmm(pj.unsqueeze(0)).squeeze(0).contiguous()
    center = torch.tensor(c2w[:3, 3].copy())      # p3d_camera.get_camera_center()
    return Camera(int(width), int(height), float(fov_x), float(fov_y), wv, pj, full, center, name, znear, zfar,
                  wv.inverse().contiguous())


def sugar_orbit_cameras(num_views: int, width: int, height: int, cx_ndc: float = 0.037, cy_ndc: float = -0.021,
                        fovx_deg: float = 60.0, radius: float = 4.0, theta_deg: float = 30.0) -> List[Camera]:
    """The orbit of ``orbit_cameras`` as SuGaR would render it: poses converted to nerfstudio's OpenGL axes and pushed
    through ``sugar_camera`` with an off-centre principal point (a calibrated COLMAP camera is never exactly centred)."""
    fovx = math.radians(fovx_deg)
    fovy = focal2fov(fov2focal(fovx, width), height)
    out = []
    for i, c2w in enumerate(orbit_c2w(radius, num_views, theta_deg)):
        gl = np.array(c2w, dtype=np.float64)
        gl[:3, 1:3] *= -1                         # OpenCV -> OpenGL: what a nerfstudio transform_matrix holds
        out.append(sugar_camera(gl, fovx, fovy, width, height, cx_ndc, cy_ndc, "{0:05d}".format(i)))
    return out


def _unit(v: np.ndarray, eps: float = 1e-10) -> np.ndarray:
    return v / (np.linalg.norm(v) + eps)


def lookat_rotation(lookat: np.ndarray, up: np.ndarray) -> np.ndarray:
    """Camera-to-world rotation with OpenCV axes (x right, y down, z forward)."""
    z = _unit(lookat)
    x = _unit(np.cross(z, up))
    y = _unit(np.cross(z, x))
    return np.array((x, y, z)).T


def orbit_positions(radius: float, num_views: int, theta_deg: float, phi_range=(0.0, 360.0)) -> np.ndarray:
    theta = np.deg2rad(np.array([theta_deg], dtype=np.float64))
    phi = np.deg2rad(np.linspace(phi_range[0], phi_range[1], num_views // len(theta) + 1)[:-1])
    theta, phi = np.meshgrid(theta, phi)
    theta, phi = theta.flatten(), phi.flatten()
    return np.stack((np.cos(theta) * np.cos(phi) * radius, np.cos(theta) * np.sin(phi) * radius,
                     np.sin(theta) * radius), axis=-1)


def orbit_c2w(radius: float, num_views: int, theta_deg: float = 30.0, center=(0.0, 0.0, 0.0),
              phi_range=(0.0, 360.0)) -> np.ndarray:
    center = np.asarray(center, dtype=np.float64)
    poses = []
    for t in orbit_positions(radius, num_views, theta_deg, phi_range) + center:
        R = lookat_rotation(center - t, np.array([0.0, 0.0, 1.0]))
        c2w = np.eye(4)
        c2w[:3, :3] = R
        c2w[:3, 3] = t
        poses.append(c2w)
    return np.stack(poses, axis=0)


def orbit_cameras(num_views: int, width: int, height: int, fovx_deg: float = 60.0, radius: float = 4.0,
                  theta_deg: float = 30.0, center=(0.0, 0.0, 0.0)) -> List[Camera]:
    """FoVx fixed, square pixels (fy = fx)."""
    fx = fov2focal(math.radians(fovx_deg), width)
    return [Camera.from_c2w(c2w, fx, fx, width, height, "{0:05d}".format(i))
            for i, c2w in enumerate(orbit_c2w(radius, num_views, theta_deg, center))]


def trajectory_dict(name: str, poses: Sequence[np.ndarray], fx: float, fy: float, cx: float, cy: float,
                    width: int, height: int) -> dict:
    """The ``custom_camera_path/<name>.json`` schema (``sample_custom_traj.py:95-106``)."""
    return {"trajectory_name": name, "camera_model": "OPENCV", "fl_x": fx, "fl_y": fy, "cx": cx, "cy": cy,
            "w": width, "h": height,
            "frames": [{"filename": "{:05d}.png".format(i), "transform_matrix": np.asarray(p).tolist()}
                       for i, p in enumerate(poses)]}


def cameras_from_trajectory(traj: dict, downscale_factor: float = 1.0) -> List[Camera]:
    """Consume a trajectory dict/JSON the way ``scene_representation.py:123-156`` does
    (frames sorted by filename; cx, cy ignored; optional integer downscale)."""
    if isinstance(traj, (str, bytes)):
        with open(traj, "r") as f:
            traj = json.load(f)
    fx, fy, w, h = traj["fl_x"], traj["fl_y"], traj["w"], traj["h"]
    if downscale_factor > 1.0:
        h, w = round(h / downscale_factor), round(w / downscale_factor)
        fx, fy = fx / downscale_factor, fy / downscale_factor
    frames = dict(sorted((fr["filename"], np.array(fr["transform_matrix"])) for fr in traj["frames"]))
    return [Camera.from_c2w(c2w, fx, fy, w, h, "{0:05d}".format(i)) for i, c2w in enumerate(frames.values())]
