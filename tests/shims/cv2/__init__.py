"""Double of the OpenCV calls on the reference's frame-output path (scene_representation.py:433-438,
sugar/gaussian_splatting/render.py:45-49): ``imwrite`` of 8-bit BGR images as PNG, ``cvtColor(RGB2BGR)``,
``applyColorMap(COLORMAP_TURBO)`` (the published turbo table, see autovfx_amd/frame_io.py)."""
import numpy as np
from PIL import Image

COLOR_RGB2BGR = 4
COLOR_BGR2RGB = 4
COLORMAP_TURBO = 20
IMREAD_ANYCOLOR = 4
IMREAD_ANYDEPTH = 2
IMREAD_UNCHANGED = -1


def cvtColor(img, code):
    assert code == COLOR_RGB2BGR
    return np.ascontiguousarray(np.asarray(img)[..., ::-1])


def applyColorMap(gray, colormap):
    assert colormap == COLORMAP_TURBO
    from autovfx_amd.frame_io import TURBO_LUT
    return np.ascontiguousarray(TURBO_LUT[np.asarray(gray, np.uint8)][..., ::-1])     # OpenCV hands out BGR


def imwrite(path, img):
    a = np.asarray(img)
    assert a.dtype == np.uint8
    if a.ndim == 3 and a.shape[2] == 3:
        a = a[..., ::-1]
    Image.fromarray(np.ascontiguousarray(a)).save(path)
    return True
