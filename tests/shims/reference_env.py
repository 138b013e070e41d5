"""Import the REFERENCE's own ``scene_representation.py`` (and everything it pulls in from its tree) in this image.

``reference_tree()`` is a context manager: inside it ``/root/reference`` (+ its two nested roots, as the reference's own
``sys.path`` hacks arrange) is importable, the doubles of this directory stand in for the third-party packages that are not
installed, import-only placeholders stand in for the ones whose functions the frame loop never calls, and -- this image's build
container has no GPU -- the few ``device="cuda"`` / ``.cuda()`` spellings in the reference's host code land on the CPU.  On exit
every module imported meanwhile is dropped again and ``sys.path`` is restored, so other tests see nothing of it.
"""
from __future__ import annotations

import contextlib
import os
import sys
import types
from unittest import mock

import torch

REF = "/root/reference"
SHIMS = os.path.dirname(os.path.abspath(__file__))
# imported by the reference at module level, never called on the frame-loop path
_PLACEHOLDERS = ("kornia", "simple_knn", "simple_knn._C", "e3nn", "imageio", "imageio.v2", "skimage", "skimage.transform", "pytorch3d",
                 "pytorch3d.renderer", "pytorch3d.renderer.cameras", "pytorch3d.transforms", "pytorch3d.ops", "pytorch3d.structures",
                 "pytorch3d.renderer.blending", "plotly", "plotly.graph_objs", "lpips", "wandb",
                 # the reference's own packages whose imports reach diffusion / inpainting / tracking models
                 "lighting", "lighting.difflight", "inpaint", "inpaint.inpaint_anything", "inpaint.retrain_utils")


class _Placeholder(types.ModuleType):
    """Any attribute is another placeholder; calling one raises (nothing on the tested path may rely on it)."""

    def __getattr__(self, key):
        if key.startswith("__"):
            raise AttributeError(key)
        child = _Placeholder(self.__name__ + "." + key)
        setattr(self, key, child)
        return child

    def __call__(self, *a, **k):
        raise RuntimeError(f"{self.__name__} is an import-only placeholder of the test harness")


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "scene_representation.py"))


def _strip_device(fn):
    return lambda *a, **k: fn(*a, **{kk: v for kk, v in k.items() if kk != "device"})


@contextlib.contextmanager
def reference_tree(cpu: bool = True):
    before_modules, before_path = dict(sys.modules), list(sys.path)
    patches = []
    try:
        tests_dir = os.path.dirname(SHIMS)
        for p in (tests_dir, SHIMS):
            if p not in sys.path:
                sys.path.insert(0, p)
        for p in (REF, os.path.join(REF, "sugar"), os.path.join(REF, "sugar", "gaussian_splatting")):
            if p not in sys.path:
                sys.path.append(p)
        for name in list(sys.modules):      # placeholders other tests registered under the doubles' names would shadow the doubles
            if name.split(".")[0] in ("torchvision", "cv2", "trimesh", "open3d", "plyfile", "kornia", "simple_knn", "scene", "utils",
                                      "gaussian_renderer", "sugar", "blender", "gaussians_utils", "rotation_utils", "scene_representation",
                                      "arguments", "opt"):
                del sys.modules[name]
        for name in _PLACEHOLDERS:
            sys.modules[name] = _Placeholder(name)
        sys.modules["kornia"].create_meshgrid = lambda H, W, normalized_coordinates=False, device=None: torch.stack(
            torch.meshgrid(torch.arange(W, dtype=torch.float32), torch.arange(H, dtype=torch.float32), indexing="xy"), -1)[None]
        if cpu:
            for fn in ("tensor", "zeros", "ones", "empty", "full", "eye"):
                patches.append(mock.patch(f"torch.{fn}", _strip_device(getattr(torch, fn))))
            patches.append(mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self))
            patches.append(mock.patch.object(torch.nn.Module, "cuda", lambda self, *a, **k: self))
            for p in patches:
                p.start()
        yield
    finally:
        for p in reversed(patches):
            p.stop()
        for name in list(sys.modules):
            if name not in before_modules:
                del sys.modules[name]
        for name, m in before_modules.items():
            sys.modules[name] = m
        sys.path[:] = before_path
