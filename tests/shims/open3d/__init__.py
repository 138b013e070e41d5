"""Double of the open3d calls of the melting branch (scene_representation.py:389-409): a ray-casting scene that answers
``compute_closest_points(points)['primitive_ids']``.  "Closest" here is by distance to the triangle's centroid (brute force): a
deterministic stand-in, the same function for the reference's loop and for the drop-in."""
import types

import numpy as np

scenes_built = 0


class _Ids:
    def __init__(self, a):
        self._a = a

    def cpu(self):
        return self

    def numpy(self):
        return self._a


class RaycastingScene:
    def __init__(self):
        global scenes_built
        scenes_built += 1
        self._centres = None

    def add_triangles(self, mesh):
        self._centres = np.asarray(mesh.triangles_center, np.float64)

    def compute_closest_points(self, points):
        p = np.asarray(points, np.float64)
        d = ((p[:, None, :] - self._centres[None, :, :]) ** 2).sum(-1)
        return {"primitive_ids": _Ids(d.argmin(axis=1).astype(np.uint32))}


class TriangleMesh:
    @staticmethod
    def from_legacy(mesh):
        return mesh


class Tensor:
    @staticmethod
    def from_numpy(a):
        return a


t = types.SimpleNamespace(geometry=types.SimpleNamespace(RaycastingScene=RaycastingScene, TriangleMesh=TriangleMesh))
core = types.SimpleNamespace(Tensor=Tensor)
