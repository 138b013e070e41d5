"""``autovfx_amd.install()``: put the MI355X render path behind an UNCHANGED AutoVFX process.

AutoVFX reaches the rasterizer through two imports (paths under the reference tree):

* ``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
  (``sugar/gaussian_splatting/gaussian_renderer/__init__.py:16``, ``sugar/sugar_scene/sugar_model.py:9``) -- served by the
  ``diff_gaussian_rasterization`` package at the root of this repository once that root is on ``sys.path``;
* ``from sugar.gaussian_splatting.gaussian_renderer import render`` (``scene_representation.py:24``,
  ``extract/extract_object.py:14``; ``from gaussian_renderer import render`` in ``sugar/gaussian_splatting/train.py:16`` /
  ``render.py:17``; ``from gaussian_splatting.gaussian_renderer import render as gs_render`` in
  ``sugar/sugar_scene/gs_model.py:7``) -- the per-frame function, whose PyTorch preparation costs more than the rasterizer.

``install()`` makes both resolve here:

1. the repository root goes to the front of ``sys.path`` (so ``diff_gaussian_rasterization`` is this one);
2. a module named ``...blend_all`` (``blender/blend_all.py``, imported at ``scene_representation.py:13``) gets its ``blend_frames``
   replaced by ``autovfx_amd.compositor.blend_frames`` (same arguments, same files in and out; PIL's resizes and the per-pixel
   composite run on the GPU);
3. a module named ``...scene_representation`` gets ``SceneRepresentation.render_from_3DGS`` (the frame loop, ``:337-447``) replaced
   by ``autovfx_amd.frame_loop.render_from_3DGS``: same arguments and files; inserted objects are loaded once instead of once per
   frame, several frames are in flight, the four files of a frame are built on the GPU;
4. a module named ``...sugar_model`` gets ``SuGaR.render_image_gaussian_rasterizer`` (two rasterizer calls over the same geometry) run with
   the binding's geometry reuse switched on for its duration: the second call costs one blend launch, same bits;
5. every module named ``...gaussian_renderer`` -- already imported or imported later (a ``sys.meta_path`` hook) -- gets its
   ``render`` replaced by ``autovfx_amd.renderer.render`` (same signature, same result dictionary; the original stays
   reachable as ``<module>.reference_render``), and every already-imported module that holds the original function under
   any name (``from ... import render [as gs_render]``) is rebound too.

Nothing else of the reference is touched: its ``GaussianModel``, cameras, scene editing and I/O run as they are.  With
autograd off, ``render`` reads the model's six raw parameter tensors and activates them inside the HIP kernels
(``gsr_forward_raw``); with autograd on it keeps the reference's structure (PyTorch activations, two rasterizer calls).

Opt-in without touching AutoVFX's sources: put ``<repo>/integration`` and ``<repo>`` on ``PYTHONPATH`` and set
``AUTOVFX_AMD_INSTALL=1``; ``integration/sitecustomize.py`` then calls ``install()`` at interpreter start (the hook itself
imports neither torch nor the HIP library until a ``gaussian_renderer`` module is actually imported).
"""
from __future__ import annotations

import importlib.abc
import importlib.util
import os
import sys
import types
from typing import Callable, List, Optional

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_TARGET_LEAF = "gaussian_renderer"
_BLEND_LEAF = "blend_all"                # blender/blend_all.py: its blend_frames() is called at scene_representation.py:232
_SCENE_LEAF = "scene_representation"     # scene_representation.py: SceneRepresentation.render_from_3DGS is the frame loop (:337-447)
_SUGAR_LEAF = "sugar_model"              # sugar/sugar_scene/sugar_model.py: SuGaR.render_image_gaussian_rasterizer calls the rasterizer twice (:2141,2174)
_installed: Optional["_RendererHook"] = None
patched_modules: List[str] = []          # names of the modules whose ``render`` was replaced (introspection / tests)
_strict = True                           # install(strict=...): may a failure to load the render path break the importing process?
_gave_up = False                         # lenient mode: the render path could not be loaded once; do not try again


def _is_target(fullname: str) -> bool:
    return any(fullname == leaf or fullname.endswith("." + leaf) for leaf in (_TARGET_LEAF, _BLEND_LEAF, _SCENE_LEAF, _SUGAR_LEAF))


def _is_blend_module(name: str) -> bool:
    return name == _BLEND_LEAF or name.endswith("." + _BLEND_LEAF)


def _is_scene_module(name: str) -> bool:
    return name == _SCENE_LEAF or name.endswith("." + _SCENE_LEAF)


def _our_frame_loop() -> Callable:
    from .frame_loop import render_from_3DGS
    return render_from_3DGS


def _patch_scene_module(module: types.ModuleType) -> None:
    """``SceneRepresentation.render_from_3DGS`` (scene_representation.py:337-447) becomes autovfx_amd.frame_loop.render_from_3DGS:
    same arguments, same directories, names and file contents; objects loaded once instead of per frame, frames in flight, file
    images built on the GPU.  The reference's method stays reachable as ``SceneRepresentation.reference_render_from_3DGS``."""
    global _gave_up
    cls = module.__dict__.get("SceneRepresentation")
    original = getattr(cls, "__dict__", {}).get("render_from_3DGS") if isinstance(cls, type) else None
    if original is None or (getattr(original, "__module__", None) or "").startswith("autovfx_amd") or _gave_up:
        return
    try:
        ours = _our_frame_loop()
    except Exception as e:
        if _strict:
            raise
        _gave_up = True
        sys.stderr.write(f"[autovfx_amd] {module.__name__}.SceneRepresentation.render_from_3DGS left as the reference's: the frame loop "
                         f"could not be loaded ({e!r})\n")
        return
    cls.reference_render_from_3DGS = original
    cls.render_from_3DGS = ours
    if module.__name__ not in patched_modules:
        patched_modules.append(module.__name__)


def _is_sugar_module(name: str) -> bool:
    return name == _SUGAR_LEAF or name.endswith("." + _SUGAR_LEAF)


def _patch_sugar_module(module: types.ModuleType) -> None:
    """``SuGaR.render_image_gaussian_rasterizer`` (sugar/sugar_scene/sugar_model.py:1960-2230, BASELINE configs[3]) rasterizes the same
    geometry twice -- colours at :2141, the per-Gaussian normals as colours at :2174 -- with nothing in between that writes to the
    positions, scales, rotations or the camera.  The method itself stays the reference's; it is merely run with the binding's
    geometry reuse switched ON for its duration (``diff_gaussian_rasterization._C.set_geometry_cache``: off by default because a
    write that bypasses PyTorch's version counters between two calls would be invisible to it -- here the code between the two calls
    is known): the second call then blends its colours over the first call's lists, one launch instead of a whole pipeline, same
    bits.  The original stays reachable as ``SuGaR.reference_render_image_gaussian_rasterizer``."""
    cls = module.__dict__.get("SuGaR")
    original = getattr(cls, "__dict__", {}).get("render_image_gaussian_rasterizer") if isinstance(cls, type) else None
    if original is None or getattr(original, "_autovfx_amd_wrapped", False):
        return

    import functools

    @functools.wraps(original)
    def render_image_gaussian_rasterizer(self, *args, **kwargs):
        from diff_gaussian_rasterization import _C
        before = _C.geometry_cache_enabled()
        _C.set_geometry_cache(True)
        try:
            return original(self, *args, **kwargs)
        finally:
            _C.set_geometry_cache(before)

    render_image_gaussian_rasterizer._autovfx_amd_wrapped = True
    cls.reference_render_image_gaussian_rasterizer = original
    cls.render_image_gaussian_rasterizer = render_image_gaussian_rasterizer
    if module.__name__ not in patched_modules:
        patched_modules.append(module.__name__)


def _our_render() -> Callable:
    from .renderer import render   # imports torch and loads libgsr_hip.so: only when a renderer module really appears
    return render


def _our_blend_frames() -> Callable:
    from .compositor import blend_frames
    return blend_frames


def _patch_renderer_module(module: types.ModuleType) -> None:
    global _gave_up
    if _is_scene_module(module.__name__):
        _patch_scene_module(module)
        return
    if _is_sugar_module(module.__name__):
        _patch_sugar_module(module)
        return
    if _is_blend_module(module.__name__):
        # the compositing step of the edit loop: ``blend_all.blend_frames(results_dir, cfg_path)`` (scene_representation.py:232) becomes
        # autovfx_amd.compositor.blend_frames -- same arguments, same files in and out, the resizes and the per-pixel composite on the GPU
        original = module.__dict__.get("blend_frames")
        if original is None or (getattr(original, "__module__", None) or "").startswith("autovfx_amd") or _gave_up:
            return
        try:
            ours = _our_blend_frames()
        except Exception as e:
            if _strict:
                raise
            _gave_up = True
            sys.stderr.write(f"[autovfx_amd] {module.__name__}.blend_frames left as the reference's: the GPU compositor could not be loaded ({e!r})\n")
            return
        module.reference_blend_frames = original
        module.blend_frames = ours
        if module.__name__ not in patched_modules:
            patched_modules.append(module.__name__)
        return
    original = module.__dict__.get("render")
    if original is None or (getattr(original, "__module__", None) or "").startswith("autovfx_amd"):
        return
    if _gave_up:
        return
    try:
        ours = _our_render()
    except Exception as e:   # torch absent, libgsr_hip.so not built / stale ABI, a GPU-less helper that inherited the environment
        if _strict:
            raise
        # The start-up hook (integration/sitecustomize.py) promised never to break the process: the reference's own render() stays
        # in place.  That is not a quiet fallback for rendering -- the reference's render() imports ``diff_gaussian_rasterization``,
        # which is this repository's package and raises when the HIP library cannot be loaded -- it only lets a process that imports
        # the renderer module without ever rendering (a data-preparation helper on a machine without a GPU) live.
        _gave_up = True
        sys.stderr.write(f"[autovfx_amd] {module.__name__}.render left as the reference's: the MI355X render path could not be loaded "
                         f"({e!r}); not retried in this process\n")
        return
    module.reference_render = original
    module.render = ours
    if module.__name__ not in patched_modules:
        patched_modules.append(module.__name__)
    # importers that already bound the original under some name (``from ... import render as gs_render``)
    for other in list(sys.modules.values()):
        d = getattr(other, "__dict__", None)
        if not isinstance(d, dict) or other is module:
            continue
        for key, value in list(d.items()):
            if value is original:
                d[key] = ours


class _PatchingLoader(importlib.abc.Loader):
    def __init__(self, inner):
        self._inner = inner

    def create_module(self, spec):
        return self._inner.create_module(spec) if hasattr(self._inner, "create_module") else None

    def exec_module(self, module):
        self._inner.exec_module(module)
        _patch_renderer_module(module)

    def __getattr__(self, name):   # get_code, get_source, is_package, ... for tools that ask the loader
        return getattr(self._inner, name)


class _RendererHook(importlib.abc.MetaPathFinder):
    """Finds ``...gaussian_renderer`` with the regular finders and wraps its loader so that ``render`` is replaced right
    after the module body ran."""

    def find_spec(self, fullname, path=None, target=None):
        if not _is_target(fullname):
            return None
        for finder in sys.meta_path:
            if finder is self or not hasattr(finder, "find_spec"):
                continue
            spec = finder.find_spec(fullname, path, target)
            if spec is not None and spec.loader is not None:
                spec.loader = _PatchingLoader(spec.loader)
                return spec
        return None


def install(path: bool = True, strict: bool = True) -> None:
    """Idempotent.  ``path=False`` leaves ``sys.path`` alone (the caller arranged for ``diff_gaussian_rasterization``).
    ``strict`` (default): a render path that cannot be loaded -- no torch, libgsr_hip.so missing or of another ABI -- raises from
    the import of the renderer module, loudly, where it happens.  ``strict=False`` is for the interpreter start-up hook
    (integration/sitecustomize.py: every Python process of the machine runs it): one line on stderr, the module keeps the
    reference's ``render``, no second attempt in that process."""
    global _installed, _strict, _gave_up
    _strict = bool(strict)
    if strict:
        _gave_up = False
    if path and (not sys.path or sys.path[0] != _REPO_ROOT):
        if _REPO_ROOT in sys.path:
            sys.path.remove(_REPO_ROOT)
        sys.path.insert(0, _REPO_ROOT)
    stale = sys.modules.get("diff_gaussian_rasterization")
    if stale is not None and not os.path.abspath(getattr(stale, "__file__", "") or "").startswith(_REPO_ROOT):
        raise RuntimeError(f"diff_gaussian_rasterization is already imported from {getattr(stale, '__file__', '?')}: call "
                           "autovfx_amd.install() before anything imports the rasterizer")
    if _installed is None:
        _installed = _RendererHook()
        sys.meta_path.insert(0, _installed)
    for name, module in list(sys.modules.items()):
        if module is not None and _is_target(name):
            _patch_renderer_module(module)


def uninstall() -> None:
    """Remove the import hook and put the reference's ``render`` back into the modules ``install`` patched (importers that
    were rebound keep what they hold; meant for tests)."""
    global _installed, _strict, _gave_up
    if _installed is not None and _installed in sys.meta_path:
        sys.meta_path.remove(_installed)
    _installed = None
    _strict, _gave_up = True, False
    for name in list(patched_modules):
        module = sys.modules.get(name)
        if module is not None and hasattr(module, "reference_render"):
            module.render = module.reference_render
        if module is not None and hasattr(module, "reference_blend_frames"):
            module.blend_frames = module.reference_blend_frames
        cls = getattr(module, "SceneRepresentation", None) if module is not None else None
        if isinstance(cls, type) and "reference_render_from_3DGS" in cls.__dict__:
            cls.render_from_3DGS = cls.reference_render_from_3DGS
            del cls.reference_render_from_3DGS
        cls = getattr(module, "SuGaR", None) if module is not None else None
        if isinstance(cls, type) and "reference_render_image_gaussian_rasterizer" in cls.__dict__:
            cls.render_image_gaussian_rasterizer = cls.reference_render_image_gaussian_rasterizer
            del cls.reference_render_image_gaussian_rasterizer
    patched_modules.clear()
