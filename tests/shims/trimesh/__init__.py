"""Double of the trimesh calls of the frame loop and gaussians_utils.py (``load_mesh`` -> ``.vertices``, ``.bounds``,
``.triangles_center``, ``.as_open3d``).  A "mesh file" of the tests is an ``np.savez`` archive with ``vertices [V,3]`` and
``faces [F,3]`` stored under whatever name the loop expects (``*.stl``, ``*.obj``, ``*.glb``)."""
import numpy as np

load_count = 0


class Mesh:
    def __init__(self, vertices, faces):
        self.vertices, self.faces = np.asarray(vertices, np.float64), np.asarray(faces, np.int64)

    @property
    def bounds(self):
        return np.stack((self.vertices.min(0), self.vertices.max(0)))

    @property
    def triangles(self):
        return self.vertices[self.faces]

    @property
    def triangles_center(self):
        return self.triangles.mean(axis=1)

    @property
    def as_open3d(self):
        return self


def load_mesh(path, **_kw):
    global load_count
    load_count += 1
    with np.load(path, allow_pickle=False) as z:
        return Mesh(z["vertices"], z["faces"])


def save_mesh(path, vertices, faces):
    with open(path, "wb") as f:
        np.savez(f, vertices=np.asarray(vertices, np.float64), faces=np.asarray(faces, np.int64))
