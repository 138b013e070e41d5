"""Test infrastructure: run the REFERENCE's own ``GaussianModel.save_ply`` / ``load_ply``
(``/root/reference/sugar/gaussian_splatting/scene/gaussian_model.py:201-266``) without the ``plyfile`` package, which this
image does not have.  ``install()`` registers a ~40-line stand-in for the four things that code touches --
``PlyData.read`` / ``PlyData([el]).write``, ``PlyElement.describe``, ``element[name]`` and ``element.properties[i].name`` --
reading and writing the binary little-endian vertex table plyfile itself produces; ``reference_gaussian_model()`` loads the
reference module from where it lies (nothing is copied).  The PLY parsing here is deliberately independent of
``autovfx_amd.gaussian_model.read_ply_vertex_table`` (numpy structured dtype built from the header)."""
import importlib.util
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

GS = "/root/reference/sugar/gaussian_splatting"


class PlyProperty:
    def __init__(self, name):
        self.name = name


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data
        self.properties = tuple(PlyProperty(n) for n in data.dtype.names)

    def __getitem__(self, key):
        return self.data[key]

    @staticmethod
    def describe(data, name):
        return PlyElement(name, data)


class PlyData:
    def __init__(self, elements):
        self.elements = list(elements)

    @staticmethod
    def read(path):
        with open(path, "rb") as f:
            assert f.readline().strip() == b"ply"
            assert f.readline().split()[:2] == [b"format", b"binary_little_endian"]
            names, count = [], None
            while True:
                line = f.readline().split()
                if line[0] == b"end_header":
                    break
                if line[0] == b"element":
                    assert line[1] == b"vertex" and count is None
                    count = int(line[2])
                if line[0] == b"property":
                    assert line[1] in (b"float", b"float32")
                    names.append(line[2].decode())
            data = np.frombuffer(f.read(), dtype=[(k, "<f4") for k in names], count=count)
        return PlyData([PlyElement("vertex", data)])

    def write(self, path):
        el = self.elements[0]
        with open(path, "wb") as f:
            f.write(b"ply\nformat binary_little_endian 1.0\n" + f"element {el.name} {len(el.data)}\n".encode())
            for k in el.data.dtype.names:
                f.write(f"property float {k}\n".encode())
            f.write(b"end_header\n")
            f.write(el.data.astype([(k, "<f4") for k in el.data.dtype.names]).tobytes())


def available() -> bool:
    return os.path.isdir(GS)


def reference_gaussian_model():
    """The reference's ``scene/gaussian_model.py`` as a module (loaded by path: its package ``__init__`` pulls in the
    COLMAP readers), with the packages it imports at load time stubbed."""
    for missing in ("plyfile", "simple_knn", "simple_knn._C", "kornia", "trimesh", "cv2", "open3d"):
        sys.modules.setdefault(missing, types.ModuleType(missing))
    sys.modules["simple_knn._C"].distCUDA2 = None
    if not hasattr(sys.modules["kornia"], "create_meshgrid"):
        sys.modules["kornia"].create_meshgrid = None
    sys.modules["plyfile"].PlyData, sys.modules["plyfile"].PlyElement = PlyData, PlyElement
    if GS not in sys.path:
        sys.path.insert(0, GS)
    spec = importlib.util.spec_from_file_location("_reference_scene_gaussian_model", os.path.join(GS, "scene", "gaussian_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.PlyData, mod.PlyElement = PlyData, PlyElement   # (another test may have stubbed `plyfile` with placeholders first)
    return mod


def on_cpu():
    """The reference's load_ply builds its tensors with device="cuda"; run it on the host."""
    real = torch.tensor
    return mock.patch("torch.tensor", lambda *a, **k: real(*a, **{kk: v for kk, v in k.items() if kk != "device"}))


FIELDS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")
