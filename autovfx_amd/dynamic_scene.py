"""Dynamic scenes: a static base cloud plus rigidly moving inserted objects, composed per frame without copying the scene.

What the reference does for every frame of an edited scene (``scene_representation.py:357-372``): ``copy.deepcopy`` of the
whole scene, ``load_ply`` of every inserted object from disk, ``transform_gaussians`` (``gaussians_utils.py:85-118``:
scale, rotate and translate the object's raw parameters about its initial centre, ~10 PyTorch launches),
``merge_two_gaussians`` (``:71-82``: six ``torch.cat`` over everything, ~0.7 GB of copies for a 3 M-Gaussian scene), and
then ``render()`` re-activates all of it.

Here the scene lives in ONE set of resident, already ACTIVATED buffers sized for the base plus every object (288 GB of
HBM hold hundreds of such scenes): the base part is written once, each object's raw parameters are loaded to the GPU
once, and per frame one kernel per placed object (``gsr_place_object``, include/gsr.h) transforms, activates and writes that
object's Gaussians at its offset.  A frame's cloud is a prefix view of the buffers -- base first, then the objects placed
in that frame, in placement order, exactly the concatenation order of the reference -- handed to the rasterizer as is.
The arithmetic is the reference's, operation by operation (``oracle/dynamic_oracle.py`` restates it; tests compare).

SH degree of a composed frame.  The reference renders every frame that has at least one placed object with DC-only colour:
``merge_two_gaussians`` returns a FRESH ``GaussianModel`` (``gaussians_utils.py:75``) whose ``active_sh_degree`` is the
constructor's 0 (``scene/gaussian_model.py:49``) -- nothing ever raises it -- and ``render()`` hands ``pc.active_sh_degree`` to
the rasterizer (``gaussian_renderer/__init__.py:111``); a frame without placements renders the deep-copied scene at its full
degree.  ``DynamicScene`` reproduces that by default (``placed_sh_degree=0``: the reference's frames, colour pop included);
``placed_sh_degree=None`` renders every frame at the base model's degree -- what one would expect, and a deliberate deviation
from upstream.
"""
from __future__ import annotations

import ctypes
import math
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

from .scenes import GaussianCloud


def matrix_to_quaternion(R) -> np.ndarray:
    """``rotation_utils.py:24-85`` for one 3x3 matrix, fp32: (w, x, y, z) of the best-conditioned candidate."""
    f32 = np.float32
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = (f32(v) for v in np.asarray(R, dtype=f32).reshape(9))
    one = f32(1.0)
    pre = np.array([one + m00 + m11 + m22, one + m00 - m11 - m22, one - m00 + m11 - m22, one - m00 - m11 + m22], dtype=f32)
    q_abs = np.where(pre > 0, np.sqrt(np.maximum(pre, f32(0))), f32(0)).astype(f32)
    sq = (q_abs * q_abs).astype(f32)
    cand = np.array([[sq[0], m21 - m12, m02 - m20, m10 - m01], [m21 - m12, sq[1], m10 + m01, m02 + m20],
                     [m02 - m20, m10 + m01, sq[2], m12 + m21], [m10 - m01, m20 + m02, m21 + m12, sq[3]]], dtype=f32)
    cand = (cand / (f32(2.0) * np.maximum(q_abs, f32(0.1)))[:, None]).astype(f32)
    return cand[int(np.argmax(q_abs))]


def placement_block(center, R, scaling: float, initial_center) -> np.ndarray:
    """The 21 floats ``gsr_place_object`` takes: centre[3], R[9] row-major, scale, initial centre[3], q_R[4], log(scale)."""
    f32 = np.float32
    R = np.asarray(R, dtype=f32).reshape(3, 3)
    return np.concatenate((np.asarray(center, f32).reshape(3), R.reshape(9), [f32(scaling)], np.asarray(initial_center, f32).reshape(3),
                           matrix_to_quaternion(R), [f32(math.log(float(scaling)))])).astype(f32)


class _ObjectCloud:
    """One inserted object, resident: raw xyz / rotation / log-scale (what the transform acts on), and the parts a rigid
    placement does not change, activated once (opacity = sigmoid, SH = cat(dc, rest))."""

    def __init__(self, model, initial_center, device):
        t = lambda a: a.detach().to(device=device, dtype=torch.float32).contiguous()
        self.xyz, self.rotation, self.log_scale = t(model._xyz), t(model._rotation), t(model._scaling)
        self.opacity = torch.sigmoid(t(model._opacity)).contiguous()
        self.shs = torch.cat((t(model._features_dc), t(model._features_rest)), dim=1).contiguous()
        self.initial_center = np.asarray(initial_center, dtype=np.float32).reshape(3)
        self.P = int(self.xyz.shape[0])


class FrameModel:
    """One composed frame with the ``GaussianModel`` getters ``render()`` reads (``gaussian_renderer/__init__.py:118-171``):
    already activated tensors, nothing is recomputed.  Valid as long as the ``GaussianCloud`` it wraps."""

    def __init__(self, cloud: GaussianCloud, min_axis: torch.Tensor):
        self._cloud, self._min_axis = cloud, min_axis
        self.active_sh_degree = self.max_sh_degree = cloud.sh_degree

    get_xyz = property(lambda self: self._cloud.means3D)
    get_scaling = property(lambda self: self._cloud.scales)
    get_rotation = property(lambda self: self._cloud.rotations)
    get_opacity = property(lambda self: self._cloud.opacities)
    get_features = property(lambda self: self._cloud.shs)
    get_minimum_axis = property(lambda self: self._min_axis)

    def get_normal(self, dir_pp_normalized=None):
        from .gaussian_model import flip_align_view
        normal_axis, _ = flip_align_view(self._min_axis, dir_pp_normalized)
        return normal_axis / normal_axis.norm(dim=1, keepdim=True)


class DynamicScene:
    """``DynamicScene(base_model, {object_id: (object_model, initial_center)})``; models carry the reference's raw parameters
    (``autovfx_amd.gaussian_model.GaussianModel`` or anything with ``_xyz, _rotation, _scaling, _opacity, _features_dc,
    _features_rest``).  ``compose(placements)`` returns the frame's ``GaussianCloud`` (activated tensors, views of the resident
    buffers: valid until the next ``compose`` into the same slot).

    ``slots`` > 1 keeps that many independent copies of the scene buffers (a 3 M-Gaussian scene is 0.7 GB of 288): a driver
    that holds several frames in flight on several HIP streams composes frame ``i`` into slot ``i % slots`` on the stream
    that renders it, so a frame's objects are never overwritten while an earlier frame on another stream still reads them."""

    def __init__(self, base, objects: Dict[str, Tuple[object, Sequence[float]]], device="cuda:0", sh_degree: Optional[int] = None,
                 slots: int = 1, placed_sh_degree: Optional[int] = 0, copies: int = 1):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("DynamicScene places objects with a HIP kernel: it needs a GPU (there is no CPU fallback)")
        t = lambda a: a.detach().to(device=self.device, dtype=torch.float32).contiguous()
        self.objects = {k: _ObjectCloud(m, c0, self.device) for k, (m, c0) in objects.items()}
        self.P_base = int(base._xyz.shape[0])
        # ``copies``: how often one object may be placed in ONE frame (the melting branch merges an object's mesh and the mesh's
        # duplicate: two subsets of the same object, scene_representation.py:395-398)
        cap = self.P_base + max(1, int(copies)) * sum(o.P for o in self.objects.values())
        M = int(base._features_dc.shape[1] + base._features_rest.shape[1])
        for k, o in self.objects.items():
            if int(o.shs.shape[1]) != M:
                raise ValueError(f"object {k!r} has {int(o.shs.shape[1])} SH coefficients, the scene {M}")
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=self.device)
        self.sh_degree = int(sh_degree if sh_degree is not None else getattr(base, "active_sh_degree", 3))
        # degree of a frame WITH placed objects: 0 = what the reference's merged model carries (module docstring); None = as the base
        self.placed_sh_degree = None if placed_sh_degree is None else int(placed_sh_degree)
        self.capacity, self.M = cap, M
        self._slots = []
        with torch.no_grad():   # the base part, once: the activations render() would redo every frame (gaussian_model.py:95-128)
            n = self.P_base
            first = (t(base._xyz), torch.exp(t(base._scaling)), torch.nn.functional.normalize(t(base._rotation)),
                     torch.sigmoid(t(base._opacity)), torch.cat((t(base._features_dc), t(base._features_rest)), dim=1))
            for _ in range(max(1, int(slots))):
                bufs = (new(cap, 3), new(cap, 3), new(cap, 4), new(cap, 1), new(cap, M, 3), new(cap, 3))
                for dst, src in zip(bufs, first):
                    dst[:n] = src
                self._slots.append(bufs)
            from .gaussian_model import get_minimum_axis
            base_axis = get_minimum_axis(first[1], first[2]).contiguous()   # general_utils.py:135-141, once for the base
            for bufs in self._slots:
                bufs[5][:n] = base_axis
        self.means3D, self.scales, self.rotations, self.opacities, self.shs, self.min_axis = self._slots[0]

    def compose(self, placements: Iterable[Tuple[str, Sequence[float], Sequence[Sequence[float]], float]], slot: int = 0) -> GaussianCloud:
        """``placements``: the objects present in this frame, in merge order, each ``(object_id, center[3], rotation[3][3],
        scaling)`` -- ``rb_transform['pos'], ['rot'], ['scale']`` of ``scene_representation.py:364-366``.  An object may be
        placed more than once (the buffers then need room for it: ``ValueError`` otherwise).

        A fifth entry selects a SUBSET of the object's Gaussians: a boolean mask of length n (numpy / torch, host or GPU) or
        an int tensor of ascending indices -- the melting branch of the frame loop (``scene_representation.py:373-421``: the
        Gaussians whose closest mesh triangle survives in the frame's melting mesh, ``orig_gaussians._xyz[mask]`` ... merged
        as they are).  That branch applies no transform: pass ``center = rotation = scaling = None`` and the subset is merged
        untouched (positions bit for bit), exactly as the reference merges it; with a transform the subset is moved like a
        whole object would be."""
        from . import _lib
        means3D, scales, rotations, opacities, shs, min_axis = self._slots[slot % len(self._slots)]
        at = self.P_base
        placements = list(placements)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            for entry in placements:
                obj_id, center, rotation, scaling = entry[:4]
                subset = self._subset_indices(entry[4], self.objects[obj_id].P) if len(entry) > 4 and entry[4] is not None else None
                o = self.objects[obj_id]
                count = o.P if subset is None else int(subset.numel())
                if at + count > self.capacity:
                    raise ValueError("the scene buffers have no room for another copy of " + repr(obj_id))
                if center is None and (rotation is not None or scaling is not None):
                    raise ValueError("an untransformed placement has center = rotation = scaling = None")
                block = None if center is None else (ctypes.c_float * 21)(*placement_block(center, rotation, scaling, o.initial_center).tolist())
                outs = (means3D[at:].data_ptr(), scales[at:].data_ptr(), rotations[at:].data_ptr(), opacities[at:].data_ptr(),
                        shs[at:].data_ptr(), min_axis[at:].data_ptr(), stream)
                if count == 0:
                    continue
                if subset is None and block is not None:
                    rc = _lib.lib.gsr_place_object(o.P, o.xyz.data_ptr(), o.rotation.data_ptr(), o.log_scale.data_ptr(), o.opacity.data_ptr(),
                                                   o.shs.data_ptr(), self.M, ctypes.byref(block), *outs)
                else:
                    if subset is None:   # the whole object, untransformed
                        subset = torch.arange(o.P, dtype=torch.int32, device=self.device)
                    # the kernel reads the list after this call returns, on THIS stream: tell the caching allocator, so that a list
                    # built on another stream (or dropped by the caller right away) is not recycled under the pending kernel
                    subset.record_stream(torch.cuda.current_stream(self.device))
                    rc = _lib.lib.gsr_place_object_subset(count, subset.data_ptr(), o.xyz.data_ptr(), o.rotation.data_ptr(),
                                                          o.log_scale.data_ptr(), o.opacity.data_ptr(), o.shs.data_ptr(), self.M,
                                                          None if block is None else ctypes.byref(block), *outs)
                if rc != 0:
                    raise RuntimeError(f"gsr_place_object failed ({rc}): {_lib.last_error()}")
                at += count
        self._last_min_axis = min_axis[:at]
        degree = self.sh_degree if (not placements or self.placed_sh_degree is None) else self.placed_sh_degree
        return GaussianCloud(means3D[:at], opacities[:at], scales[:at], rotations[:at], shs[:at], None, degree)

    def _subset_indices(self, sel, n: int) -> torch.Tensor:
        """A mask or index list -> ascending int32 indices on the scene's device."""
        t = torch.as_tensor(sel)
        if t.dtype == torch.bool:
            if t.numel() != n:
                raise ValueError(f"subset mask has {t.numel()} entries, the object {n} Gaussians")
            t = torch.nonzero(t.reshape(-1), as_tuple=False).reshape(-1)   # (a GPU mask costs one host read here: its count sizes the frame)
        elif t.numel():
            # An index >= n (or a negative one, a huge offset once cast to uint32) would be an out-of-bounds device read in
            # gsr_place_object_subset.  Host lists are checked here every time; a list that already lives on the GPU -- the fast path
            # of a caller that matched meshes once and keeps the per-frame lists resident -- is checked ONCE (one host read of its
            # min and max) and remembered by (tensor object, version counter, length): the frames after the first pay nothing,
            # and an in-place edit or a new tensor is checked again.
            # The memory is the TENSOR OBJECT's (a weak reference: an address alone could be a recycled allocation).
            import weakref
            seen = self.__dict__.setdefault("_checked_subsets", {})
            entry = seen.get(id(t)) if t.is_cuda else None
            if entry is None or entry[0]() is not t or entry[1] != (t._version, t.numel(), int(n)):
                if int(t.min()) < 0 or int(t.max()) >= n:
                    raise ValueError("subset indices out of range")
                if t.is_cuda:
                    key = id(t)
                    seen[key] = (weakref.ref(t, lambda _r, key=key, seen=seen: seen.pop(key, None)), (t._version, t.numel(), int(n)))
        return t.to(device=self.device, dtype=torch.int32).contiguous()

    def compose_model(self, placements, slot: int = 0) -> FrameModel:
        """``compose`` for callers of ``render()`` (``autovfx_amd.renderer.render`` or the reference's own): the frame as an
        object with the ``GaussianModel`` getters -- RGBA, depth, normal and pseudo-normal maps of a moving scene come out of
        ``render(view, scene.compose_model(placements), pipe, bg)`` as they do for a static one."""
        cloud = self.compose(placements, slot)
        return FrameModel(cloud, self._last_min_axis)
