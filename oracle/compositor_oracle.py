"""numpy restatement of the per-frame compositing in ``blender/blend_all.py::blend_frames`` (``:166-346``).

TEST INFRASTRUCTURE ONLY.  The reference does file discovery, PNG / EXR / npy loading and PIL resizing
around this math (``:124-234``); what is restated here is the arithmetic that turns the loaded, equally sized
layers of ONE frame into the output frame (``:236-300,341-343``), in the reference's order and dtypes
(uint8 layers promoted to float32, depths float32).  ``tests/test_compositor.py`` pins it against the
reference's own ``blend_frames`` executed on synthetic layers with its loaders monkey-patched.

Layers (all H x W): ``bg_c`` background RGBA from the 3DGS render; ``o_c, o_d`` Blender object RGBA / depth;
``s_c, s_d`` shadow-catcher pass; ``o_s_c`` object + shadow-catcher pass; optional ``o_gs_c, o_gs_d`` (3DGS
objects re-rendered by Blender), ``s_f_c, s_f_d`` (smoke / fire), ``s_f_c_pre`` (premultiplied fire).
The GS depth map ``bg_d`` is loaded by the reference but never used.
"""
from __future__ import annotations

import numpy as np


def smoke_depth_fill(s_f_c, s_f_d, s_f_d_pre=None):
    """``blend_all.py:207-215``: where smoke has alpha, its depth becomes the layer's 0.001-th percentile."""
    mask = (s_f_c[..., 3] / 255.) > 0.0
    s_f_d = s_f_d.copy()
    s_f_d[mask] = np.percentile(s_f_d, 0.001)
    if s_f_d_pre is not None:
        s_f_d_pre = s_f_d_pre.copy()
        s_f_d_pre[mask] = np.percentile(s_f_d_pre, 0.001)
    return s_f_d, s_f_d_pre


def composite_frame(bg_c, o_c, o_d, s_c, s_d, o_s_c, o_gs_c=None, o_gs_d=None, s_f_c=None, s_f_d=None,
                    s_f_c_pre=None) -> np.ndarray:
    has_3dgs, has_smoke, has_fire = o_gs_c is not None, s_f_c is not None, s_f_c_pre is not None
    bg_c, o_c, s_c, o_s_c = (a.astype(np.float32) for a in (bg_c, o_c, s_c, o_s_c))
    if has_3dgs:
        o_gs_c = o_gs_c.astype(np.float32)
    if has_smoke:
        s_f_c = s_f_c.astype(np.float32)
        if has_fire:
            s_f_c_pre = s_f_c_pre.astype(np.float32)
    frame = bg_c.copy()

    # step 1: shadows onto the background (:241-280)
    if has_3dgs:
        depth_mask = s_d <= o_gs_d
        obj_3dgs_alpha = o_gs_c[..., 3] / 255.
        non_obj_3dgs_alpha = 1. - obj_3dgs_alpha
        non_obj_3dgs_alpha[depth_mask] = 1.0
    obj_alpha = o_c[..., 3] / 255.
    depth_mask = o_d <= s_d
    if has_smoke or has_fire:
        obj_alpha_smoke = s_f_c[..., 3] / 255.
        depth_mask_smoke = s_f_d <= s_d
        obj_alpha = np.maximum(obj_alpha, obj_alpha_smoke)
        depth_mask = np.logical_or(depth_mask, depth_mask_smoke)
    obj_mask = obj_alpha > 0.0
    mask = np.logical_and(obj_mask, depth_mask)
    obj_alpha[~mask] = 0.0
    non_object_alpha = 1. - obj_alpha
    if has_3dgs:
        front = o_gs_d <= o_d
        obj_alpha[front] *= non_obj_3dgs_alpha[front]
    fg_alpha = o_s_c[..., 3] / 255.
    shadow_catcher_alpha = non_object_alpha * fg_alpha * non_obj_3dgs_alpha if has_3dgs else non_object_alpha * fg_alpha
    shadow_catcher_mask = shadow_catcher_alpha > 0.0
    color_diff = np.ones_like(o_c)
    color_diff[shadow_catcher_mask, 0:3] = o_s_c[shadow_catcher_mask, :3] / (s_c[shadow_catcher_mask, :3] + 1e-6)
    color_diff = np.clip(color_diff, 0, 1)
    shadow_mask = np.logical_not(np.all(np.abs(color_diff - 1) < 0.01, axis=-1))
    m = shadow_mask
    frame[m] = frame[m] * color_diff[m] * shadow_catcher_alpha[m, None] + frame[m] * (1 - shadow_catcher_alpha[m, None])

    # step 2: objects (and fire) over the shadowed background (:285-291)
    frame_tmp = frame.copy()
    m = np.logical_and(obj_mask, depth_mask)
    frame[:, :, :3][m] = o_c[:, :, :3][m] * obj_alpha[m, None] + frame_tmp[:, :, :3][m] * (1 - obj_alpha[m, None])
    if has_fire:
        m = depth_mask_smoke
        frame[:, :, :3][m] = s_f_c_pre[:, :, :3][m] + frame_tmp[:, :, :3][m] * (1 - obj_alpha_smoke[m, None])
    return np.clip(frame, 0, 255).astype(np.uint8)
