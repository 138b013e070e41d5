"""The frame loop of an edited scene at render rate: a drop-in for ``SceneRepresentation.render_from_3DGS``
(``/root/reference/scene_representation.py:337-447``) and the loop ``scripts/render_trajectory.py`` runs.

What the reference does per frame, on one host thread, in order: (rigid-body and melting scenes only) ``copy.deepcopy`` of the
whole scene, ``load_gaussians`` of every inserted object FROM DISK, ``trimesh`` / ``open3d`` look-ups of things that do not
change between frames (the object's mesh centre; the original mesh, its ray-casting scene and the closest triangle of every
Gaussian), ``transform_gaussians``, ``merge_two_gaussians`` (``:357-423``); one blocking ``render()`` (``:424``); then
``torchvision.utils.save_image`` + ``.cpu().numpy()`` + ``np.save`` + two ``cv2.imwrite`` (``:425-438``) -- about 100 ms of host
time per 960x540 frame around a 1 ms render.

Here:
* the scene is loaded once per call (``self.load_scene()``, as the reference: an edit may have replaced the checkpoint);
* everything that does not depend on the frame is done ONCE per object -- its PLY, its mesh centre, (melting) the ray-casting
  scene and the Gaussians' closest triangles -- with the reference module's own functions (``load_gaussians``,
  ``get_center_of_mesh_2``, ``trimesh``, ``o3d`` are taken from the module that defines the scene class: nothing of them is
  restated here);
* a frame with placed objects is composed by ``DynamicScene`` -- one HIP kernel per placed object into resident, activated buffers
  (``gsr_place_object[_subset]``), bit-identical to transform -> merge -> activate; a frame without any renders the scene's own
  model, at its own SH degree, as the reference's deep copy does;
* ``streams`` frames are in flight from this one host thread (``render_begin`` / ``finish``: a frame's projection and depth sort
  are queued before the host waits for an older frame's pair count);
* the four files of a frame are built as FILE IMAGES on the GPU and leave through one device-to-host copy; host threads only
  ``write()`` (``GpuFrameWriter``); same directories, same names, same pixels / ``.npy`` bytes for every reader;
* when ``torch.distributed`` is initialised, rank r renders and writes frames r, r + N, ... (frames are the
  independent unit; the contract is the files, so there is no gather -- one barrier at the end so that whoever continues finds all
  files on disk).

No fallback: without the HIP library, or with a model that is not on the GPU, this raises; it never routes a frame through the
reference's loop behind the caller's back.  ``<class>.reference_render_from_3DGS`` stays reachable.
"""
from __future__ import annotations

import copy
import glob
import os
import sys
from collections import deque
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

DEFAULT_STREAMS = int(os.environ.get("AUTOVFX_AMD_LOOP_STREAMS", "3"))
DEFAULT_WRITER_THREADS = int(os.environ.get("AUTOVFX_AMD_LOOP_WRITERS", "4"))
LAST_LOOP_STATS: dict = {}     # AUTOVFX_AMD_LOOP_STATS=1: host seconds of the last render_frames call spent in begin / finish (waiting) / submit

# The three things the loop is made of.  They are module attributes so that the CPU test of the drop-in (no GPU there) can put
# doubles in their place EXPLICITLY; the product never rebinds them and there is no automatic choice between them.
def _render_begin(view, model, pipe, bg):
    from . import renderer
    return renderer.render_begin(view, model, pipe, bg)


def _render(view, model, pipe, bg):
    from . import renderer
    return renderer.render(view, model, pipe, bg)


def _make_writer(out_dir: str, writer_threads: int, slots: int):
    from . import frame_io
    return frame_io.GpuFrameWriter(out_dir, workers=writer_threads, slots=slots)


def _make_dynamic_scene(base, objects, device, sh_degree, slots, copies):
    from .dynamic_scene import DynamicScene
    return DynamicScene(base, objects, device=device, sh_degree=sh_degree, slots=slots, placed_sh_degree=0, copies=copies)


def _side_streams(device, count):
    from .frame_parallel import side_streams
    return side_streams(device, count)


def png_mode() -> str:
    """How the PNGs of the loop are encoded: ``stored`` (deflate stored blocks) or ``deflate`` (compressed on the GPU)."""
    from . import frame_io
    return frame_io.png_mode()


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size(), dist
    except Exception:
        pass
    return 0, 1, None


def render_frames(views: Sequence, names: Sequence[str], model_for_frame: Callable[[int, int], object], out_dir: str, pipe, bg,
                  frame_ids: Optional[Iterable[int]] = None, streams: int = DEFAULT_STREAMS,
                  writer_threads: int = DEFAULT_WRITER_THREADS, progress: Optional[Callable] = None) -> int:
    """Render ``views[i]`` of the frames ``frame_ids`` (default: all) and write each one's four files under ``out_dir`` as
    ``images/<names[i]>.png``, ``depth/<names[i]>.npy``, ``depth/<names[i]>.png``, ``normal/<names[i]>.png``.
    ``model_for_frame(i, slot)`` returns what ``render()`` takes as ``pc`` for frame ``i``; ``slot`` (0 .. streams-1) tells a
    composer with per-slot buffers which copy this frame may overwrite.  Returns the number of frames written.  The caller holds
    ``torch.no_grad()``."""
    ids = list(range(len(views))) if frame_ids is None else list(frame_ids)
    S = max(1, int(streams))
    it = ids if progress is None else progress(ids)
    with _make_writer(out_dir, writer_threads, slots=max(8, 2 * S)) as writer:
        if S == 1:
            for i in it:
                writer.submit(names[i], _render(views[i], model_for_frame(i, 0), pipe, bg))
            return len(ids)
        device = bg.device
        side, q = _side_streams(device, S), deque()
        for st in side:                # what the caller queued on its stream (the model's upload, the scene buffers) comes first
            st.wait_stream(torch.cuda.current_stream(device))

        stats = LAST_LOOP_STATS if os.environ.get("AUTOVFX_AMD_LOOP_STATS") else None
        if stats is not None:          # where the host thread's time goes: begin (compose + first half), finish (wait + second half), submit (files)
            import time
            clock = time.perf_counter
            stats.clear()
            stats.update(frames=len(ids), streams=S, begin_s=0.0, finish_s=0.0, submit_s=0.0, loop_s=0.0)
            t_loop = clock()

        def finish_oldest():
            st, name, pending = q.popleft()
            with torch.cuda.stream(st):
                if stats is None:
                    writer.submit(name, pending.finish())
                else:
                    t0 = clock()
                    result = pending.finish()
                    t1 = clock()
                    writer.submit(name, result)
                    stats["finish_s"] += t1 - t0
                    stats["submit_s"] += clock() - t1

        # (One host thread on purpose.  Handing the file kernels of a frame to a second thread was tried: its Python parts take the
        # interpreter lock from this one for milliseconds at a time -- profiles/r06_loop_filer_thread_ab.txt.)
        split = True       # render() in two halves; a model / pipeline it cannot split gets one blocking render() per frame instead
        for k, i in enumerate(it):
            while len(q) == S:
                finish_oldest()
            st = side[k % S]
            with torch.cuda.stream(st):
                t0 = clock() if stats is not None else 0.0
                model = model_for_frame(i, k % S)
                pending = None
                if split:
                    try:
                        pending = _render_begin(views[i], model, pipe, bg)
                    except RuntimeError as e:
                        if "render_begin needs" not in str(e):
                            raise
                        split = False          # (same kernels, the same library: only the frames no longer overlap)
                if pending is not None:
                    q.append((st, names[i], pending))
                else:
                    writer.submit(names[i], _render(views[i], model, pipe, bg))
                if stats is not None:
                    stats["begin_s"] += clock() - t0
        while q:
            finish_oldest()
        if stats is not None:
            stats["loop_s"] = clock() - t_loop
        for st in side:
            torch.cuda.current_stream(device).wait_stream(st)
    return len(ids)


# ------------------------------------------------------------------------------------------------------------------------------------
# what a frame of an edited scene consists of (scene_representation.py:357-423), worked out once per call instead of once per frame
# ------------------------------------------------------------------------------------------------------------------------------------
class _StaticPlan:
    """``all_gaussians = self.gaussians`` (``:422-423``)."""
    def __init__(self, scene):
        self.model = scene.gaussians

    def model_for_frame(self, idx, slot):
        return self.model


def _object_info(scene, obj_id):
    return [obj for obj in scene.blender_cfg['insert_object_info'] if obj['object_id'] == obj_id][0]     # (:363, :380-383)


def _object_gaussians_path(obj_info):
    return os.path.join('/'.join(obj_info['object_path'].split('/')[:-2]), 'object_gaussians.ply')      # (:364, :385)


class _RigidBodyPlan:
    """``self.rb_transform_info`` is set (``:357-372``): per frame, every object that has an entry for frame ``idx + 1`` (three
    digits, 1-based) is transformed about its mesh centre and merged behind the scene, in the dictionary's order."""

    def __init__(self, scene, mod, num_frames, streams):
        self.scene, self.info = scene, scene.rb_transform_info
        keys = {"{0:03d}".format(i + 1) for i in range(num_frames)}
        objects = {}
        for obj_id, per_frame in self.info.items():
            if not (keys & set(per_frame)):
                continue       # never placed in this trajectory: the reference never looks it up either
            obj_info = _object_info(scene, obj_id)
            gaussians = mod.load_gaussians(_object_gaussians_path(obj_info), scene.hparams.max_sh_degree - 1)    # once, not per frame
            initial_center = np.asarray(mod.get_center_of_mesh_2(obj_info['object_path']))                       # once, not per frame
            objects[obj_id] = (gaussians, initial_center)
        self.dynamic = None
        if objects:
            base = scene.gaussians
            self.dynamic = _make_dynamic_scene(base, objects, base._xyz.device, int(base.active_sh_degree), max(1, streams), 1)

    def model_for_frame(self, idx, slot):
        key = "{0:03d}".format(idx + 1)        # frame index starts from 001 (:361-362)
        placed = [(obj_id, t[key]['pos'], t[key]['rot'], t[key]['scale']) for obj_id, t in self.info.items() if key in t]
        if not placed:
            return self.scene.gaussians        # the deep copy of the untouched scene, at the scene's own SH degree (:358)
        return self.dynamic.compose_model(placed, slot=slot)


class _MeltingPlan:
    """``<blender_cache_dir>/<output_dir_name>/melting_meshes`` exists (``:373-421``): per frame and object, the Gaussians whose
    closest triangle of the ORIGINAL mesh is also the closest triangle of some face centre of the frame's melting mesh
    (``NNN_obj.stl``, then ``NNN_obj_dup.stl``) are merged behind the scene, untransformed.  The original mesh, its ray-casting
    scene and the Gaussians' triangle ids do not depend on the frame: they are computed once per object here (the reference
    recomputes them every frame), with the reference module's own ``trimesh`` / ``o3d``."""

    def __init__(self, scene, mod, mesh_output_dir, streams):
        self.scene, self.mod, self.objects = scene, mod, []
        o3d, trimesh = mod.o3d, mod.trimesh
        models = {}
        for obj_id in sorted(os.listdir(mesh_output_dir)):                                               # (:377)
            obj_info = _object_info(scene, obj_id)
            orig_mesh_path = obj_info['object_path']
            orig_mesh = trimesh.load_mesh(orig_mesh_path)
            orig_gaussians = mod.load_gaussians(_object_gaussians_path(obj_info), scene.hparams.max_sh_degree - 1)
            ray_scene = o3d.t.geometry.RaycastingScene()
            ray_scene.add_triangles(o3d.t.geometry.TriangleMesh.from_legacy(orig_mesh.as_open3d))
            xyz = orig_gaussians._xyz.detach().cpu().numpy()
            ids = ray_scene.compute_closest_points(o3d.core.Tensor.from_numpy(xyz.astype(np.float32)))['primitive_ids'].cpu().numpy()
            self.objects.append((obj_id, os.path.join(mesh_output_dir, obj_id), ray_scene, ids))
            models[obj_id] = (orig_gaussians, (0.0, 0.0, 0.0))
        self.dynamic = None
        if models:
            base = scene.gaussians
            # an object can be merged twice in one frame (its mesh and the mesh's duplicate): room for two copies of each
            self.dynamic = _make_dynamic_scene(base, models, base._xyz.device, int(base.active_sh_degree), max(1, streams), 2)

    def model_for_frame(self, idx, slot):
        o3d, trimesh = self.mod.o3d, self.mod.trimesh
        placed = []
        for obj_id, melting_mesh_dir, ray_scene, triangle_ids_from_gaussians in self.objects:
            for path in (os.path.join(melting_mesh_dir, '{0:03d}_obj.stl'.format(idx + 1)),
                         os.path.join(melting_mesh_dir, '{0:03d}_obj_dup.stl'.format(idx + 1))):            # (:395-398)
                if not os.path.exists(path):
                    continue
                melting_mesh = trimesh.load_mesh(path)
                centres = np.array(melting_mesh.triangles_center).astype(np.float32)
                ids = ray_scene.compute_closest_points(o3d.core.Tensor.from_numpy(centres))['primitive_ids'].cpu().numpy()
                mask = np.isin(triangle_ids_from_gaussians, ids)                                          # (:411)
                placed.append((obj_id, None, None, None, mask))
        if not placed:
            return self.scene.gaussians
        return self.dynamic.compose_model(placed, slot=slot)


def _plan(scene, mod, num_frames, streams):
    if scene.rb_transform_info is not None:                                                               # (:357)
        return _RigidBodyPlan(scene, mod, num_frames, streams)
    mesh_output_dir = os.path.join(scene.blender_cache_dir, scene.hparams.blender_output_dir_name, 'melting_meshes')
    if os.path.exists(mesh_output_dir):                                                                   # (:373)
        return _MeltingPlan(scene, mod, mesh_output_dir, streams)
    return _StaticPlan(scene)


def render_from_3DGS(self, render_video=False, post_rendering=False):
    """``SceneRepresentation.render_from_3DGS`` (``scene_representation.py:337-447``): same arguments, same directories and file
    names, same pixels and ``.npy`` bytes; see the module docstring for what is different underneath."""
    mod = sys.modules[type(self).__module__]           # the reference's module: its helpers and host libraries are used as they are

    self.load_scene()  # reload the scene to get the latest gaussians (:339)

    camera_views = self.cameras['cameras']
    if post_rendering and self.hparams.render_type == 'SINGLE_VIEW':                                      # (:343-346)
        camera_views = [copy.deepcopy(self.cameras['cameras'][self.anchor_frame_idx]) for _ in range(self.total_frames)]
        for cam_idx, view in enumerate(camera_views):
            camera_views[cam_idx].image_name = '{0:05d}'.format(cam_idx)

    render_path = os.path.join(self.traj_results_dir, "images")
    depth_path = os.path.join(self.traj_results_dir, "depth")
    normal_path = os.path.join(self.traj_results_dir, "normal")
    for p in (render_path, depth_path, normal_path):
        os.makedirs(p, exist_ok=True)

    rank, world, dist = _rank_world()
    streams = DEFAULT_STREAMS
    names = [view.image_name for view in camera_views]
    import time
    with torch.no_grad():
        t0 = time.perf_counter()
        plan = _plan(self, mod, len(camera_views), streams)
        t1 = time.perf_counter()
        mine = list(range(rank, len(camera_views), world))
        tqdm = getattr(mod, "tqdm", None)
        progress = (lambda ids: tqdm(ids, desc="Rendering progress")) if tqdm is not None else None
        render_frames(camera_views, names, plan.model_for_frame, self.traj_results_dir, self.pipe, self.background,
                      frame_ids=mine, streams=streams, progress=progress)
        if os.environ.get("AUTOVFX_AMD_LOOP_STATS"):
            LAST_LOOP_STATS.update(plan_s=t1 - t0, render_frames_s=time.perf_counter() - t1)
    if dist is not None and world > 1:
        dist.barrier()                 # every rank's files are on disk before anyone reads the directory

    # generate video from frames (:440-447), with the reference's own function; one rank does it
    if render_video and rank == 0:
        for path, name in ((render_path, 'render_rgb.mp4'), (depth_path, 'render_depth.mp4'), (normal_path, 'render_normal.mp4')):
            frames = sorted(glob.glob(os.path.join(path, '*.png')))
            mod.generate_video_from_frames(frames, os.path.join(self.traj_results_dir, name), fps=15)
