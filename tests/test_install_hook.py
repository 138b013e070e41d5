"""autovfx_amd.install(): the import hook that puts this package's render() behind an unchanged AutoVFX process, and the
gate that decides whether a model may be rendered from its raw parameter tensors.  CPU only: nothing here launches a kernel."""
import importlib
import os
import subprocess
import sys
import textwrap
import types

import pytest
import torch

import autovfx_amd
from autovfx_amd import hook, renderer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GS = "/root/reference/sugar/gaussian_splatting"


@pytest.fixture
def fake_autovfx(tmp_path, monkeypatch):
    """A miniature of the reference's package layout: sugar/gaussian_splatting/gaussian_renderer with a render(), and the
    three import spellings the reference uses for it."""
    mine = lambda n: n.split(".")[0] in ("sugar", "gaussian_renderer", "scene_representation_like", "gs_model_like")
    parked = {n: sys.modules.pop(n) for n in [n for n in sys.modules if mine(n)]}   # (other tests stub or import these names)
    pkg = tmp_path / "sugar" / "gaussian_splatting" / "gaussian_renderer"
    pkg.mkdir(parents=True)
    (tmp_path / "sugar" / "__init__.py").write_text("")
    (tmp_path / "sugar" / "gaussian_splatting" / "__init__.py").write_text("")
    (pkg / "__init__.py").write_text("def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):\n"
                                     "    return 'reference'\n\ndef get_ray_directions():\n    return 'kept'\n")
    (tmp_path / "scene_representation_like.py").write_text("from sugar.gaussian_splatting.gaussian_renderer import render\n")
    (tmp_path / "gs_model_like.py").write_text("from sugar.gaussian_splatting.gaussian_renderer import render as gs_render\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    monkeypatch.syspath_prepend(str(tmp_path / "sugar" / "gaussian_splatting"))   # `from gaussian_renderer import render`
    yield tmp_path
    autovfx_amd.uninstall()
    for name in [n for n in sys.modules if mine(n)]:
        del sys.modules[name]
    sys.modules.update(parked)


def test_install_before_import_patches_the_renderer(fake_autovfx):
    autovfx_amd.install()
    assert sys.path[0] == ROOT
    caller = importlib.import_module("scene_representation_like")
    assert caller.render is renderer.render
    module = sys.modules["sugar.gaussian_splatting.gaussian_renderer"]
    assert module.render is renderer.render and module.reference_render(None, None, None, None) == "reference"
    assert module.get_ray_directions() == "kept"
    short = importlib.import_module("gaussian_renderer")          # the spelling of sugar/gaussian_splatting/train.py:16
    assert short.render is renderer.render
    assert hook.patched_modules == ["sugar.gaussian_splatting.gaussian_renderer", "gaussian_renderer"]


def test_install_after_import_rebinds_existing_importers(fake_autovfx):
    a = importlib.import_module("scene_representation_like")
    b = importlib.import_module("gs_model_like")
    assert a.render(None, None, None, None) == "reference"
    autovfx_amd.install()
    autovfx_amd.install()   # idempotent
    assert a.render is renderer.render and b.gs_render is renderer.render
    assert sum(1 for f in sys.meta_path if isinstance(f, hook._RendererHook)) == 1
    autovfx_amd.uninstall()
    assert sys.modules["sugar.gaussian_splatting.gaussian_renderer"].render(None, None, None, None) == "reference"
    assert not any(isinstance(f, hook._RendererHook) for f in sys.meta_path)


def test_install_patches_blend_all_blend_frames(fake_autovfx, monkeypatch):
    """``from blender import blend_all`` (scene_representation.py:13) ... ``blend_all.blend_frames(dir, cfg)`` (:232): after install()
    the attribute is this package's drop-in, the reference's stays reachable, uninstall() puts it back."""
    pkg = fake_autovfx / "blender"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("")
    (pkg / "blend_all.py").write_text("def blend_frames(blend_results_dir, input_config_path=None):\n    return 'reference'\n")
    for n in [n for n in sys.modules if n.split(".")[0] == "blender"]:
        monkeypatch.delitem(sys.modules, n)
    autovfx_amd.install()
    from autovfx_amd import compositor
    mod = importlib.import_module("blender.blend_all")
    assert mod.blend_frames is compositor.blend_frames and mod.reference_blend_frames("x") == "reference"
    assert "blender.blend_all" in hook.patched_modules
    autovfx_amd.uninstall()
    assert mod.blend_frames("x") == "reference"
    for n in [n for n in sys.modules if n.split(".")[0] == "blender"]:
        del sys.modules[n]


def test_install_refuses_a_foreign_rasterizer_already_imported(fake_autovfx, monkeypatch):
    foreign = types.ModuleType("diff_gaussian_rasterization")
    foreign.__file__ = "/somewhere/site-packages/diff_gaussian_rasterization/__init__.py"
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", foreign)
    with pytest.raises(RuntimeError, match="already imported"):
        autovfx_amd.install()


def test_sitecustomize_opt_in(fake_autovfx):
    """AUTOVFX_AMD_INSTALL=1 with integration/ on PYTHONPATH installs the hook at interpreter start without importing torch;
    without the variable nothing happens."""
    code = textwrap.dedent("""
        import sys
        early = 'torch' in sys.modules
        from sugar.gaussian_splatting.gaussian_renderer import render
        print(render.__module__, early)
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "integration"), ROOT, str(fake_autovfx)]))
    env.pop("AUTOVFX_AMD_INSTALL", None)
    off = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert off.returncode == 0 and off.stdout.split() == ["sugar.gaussian_splatting.gaussian_renderer", "False"], off.stderr
    on = subprocess.run([sys.executable, "-c", code], env=dict(env, AUTOVFX_AMD_INSTALL="1"), capture_output=True, text=True, timeout=300)
    assert on.returncode == 0 and on.stdout.split() == ["autovfx_amd.renderer", "False"], on.stderr


def test_start_up_hook_survives_a_render_path_that_cannot_be_loaded(fake_autovfx):
    """under AUTOVFX_AMD_INSTALL=1 every Python process of the machine runs the hook; one in which this package's
    render path cannot be imported (here: no torch -- the HIP library itself is only opened at the first render call, which then
    fails loudly) must still be able to IMPORT the reference's renderer module -- one stderr line, ``render`` stays the
    reference's, no retry at the next import.  An explicit ``autovfx_amd.install()`` stays strict: the same situation raises at
    the import."""
    code = textwrap.dedent("""
        import sys
        sys.modules['torch'] = None          # "import torch" raises from here on
        from sugar.gaussian_splatting.gaussian_renderer import render
        import gaussian_renderer
        print(render.__module__, gaussian_renderer.render.__module__)
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "integration"), ROOT, str(fake_autovfx),
                                                       str(fake_autovfx / "sugar" / "gaussian_splatting")]),
               GSR_LIB="/nonexistent/libgsr_hip.so", AUTOVFX_AMD_INSTALL="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.split() == ["sugar.gaussian_splatting.gaussian_renderer", "gaussian_renderer"]
    assert r.stderr.count("could not be loaded") == 1, r.stderr       # said once, not per module
    strict = textwrap.dedent("""
        import sys
        import autovfx_amd
        autovfx_amd.install()
        sys.modules['torch'] = None
        try:
            from sugar.gaussian_splatting.gaussian_renderer import render
        except Exception as e:
            print("raised", type(e).__name__)
    """)
    env.pop("AUTOVFX_AMD_INSTALL")
    r = subprocess.run([sys.executable, "-c", strict], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.split()[0] == "raised", (r.stdout, r.stderr)


class _Model:
    def __init__(self, P=5, M=16, device="cpu"):
        self._xyz = torch.zeros(P, 3, device=device)
        self._scaling = torch.zeros(P, 3, device=device)
        self._rotation = torch.ones(P, 4, device=device)
        self._opacity = torch.zeros(P, 1, device=device)
        self._features_dc = torch.zeros(P, 1, 3, device=device)
        self._features_rest = torch.zeros(P, M - 1, 3, device=device)


def test_raw_parameter_gate(monkeypatch):
    m = _Model()
    assert renderer.raw_parameters(m) is None                      # CPU tensors: there is no CPU path
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))   # pretend: only the gate's logic runs
    assert renderer.raw_parameters(m) is not None and len(renderer.raw_parameters(m)) == 6
    for attr, fn in renderer._RAW_ACTIVATIONS:                     # the reference keeps its activations as attributes
        setattr(m, attr, fn)
    assert renderer.raw_parameters(m) is not None
    m.opacity_activation = torch.tanh
    assert renderer.raw_parameters(m) is None
    m.opacity_activation = torch.sigmoid
    m.gsr_raw_parameters = False
    assert renderer.raw_parameters(m) is None
    del m.gsr_raw_parameters
    m._features_dc = m._features_dc.reshape(-1, 3)
    assert renderer.raw_parameters(m) is None
    m = _Model()
    m._scaling = m._scaling.double()
    assert renderer.raw_parameters(m) is None
    m = _Model()
    del m._features_rest
    assert renderer.raw_parameters(m) is None
    assert renderer.raw_parameters(_Model(M=1)) is not None        # degree 0 only: an empty [P,0,3] rest tensor


@pytest.mark.skipif(not os.path.isdir(GS), reason="reference tree not mounted")
def test_a_subclass_that_overrides_one_getter_is_not_rendered_from_its_raw_tensors(monkeypatch):
    """render() on the raw path calls no getter; a subclass with, say, a masked ``get_opacity`` would be rendered
    (and differentiated) from ``_opacity`` as if the override were not there.  The gate sends such a model down the getters path."""
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))   # (the gate wants GPU tensors; none is touched)
    from test_raw_gpu import ReferenceShapedModel
    z = lambda *s: torch.zeros(*s)
    args = (3, z(5, 3), z(5, 3), torch.ones(5, 4), z(5, 1), z(5, 1, 3), z(5, 15, 3))
    stock = ReferenceShapedModel(*args)
    assert renderer.raw_parameters(stock) is not None

    class Masked(ReferenceShapedModel):
        @property
        def get_opacity(self):
            return torch.sigmoid(self._opacity) * 0.5

    assert renderer.raw_parameters(Masked(*args)) is None

    class Plain(ReferenceShapedModel):       # a subclass that adds things but leaves the getters alone keeps the raw path
        def extra(self):
            return 1

    assert renderer.raw_parameters(Plain(*args)) is not None

    class Insists(Masked):
        gsr_raw_parameters = True

    assert renderer.raw_parameters(Insists(*args)) is not None


def test_the_references_own_gaussian_model_passes_the_gate(monkeypatch):
    """The class AutoVFX instantiates (scene/gaussian_model.py), imported unchanged: its attribute names, shapes and
    activation functions are what raw_parameters() looks for."""
    from test_render_mirror import _import_reference
    ref = _import_reference("scene.gaussian_model")
    r = ref.GaussianModel(3)
    src = _Model(P=7)
    for k, v in vars(src).items():
        setattr(r, k, v)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True))
    got = renderer.raw_parameters(r)
    assert got is not None and got[0] is r._xyz and got[5] is r._features_rest


@pytest.mark.skipif(not os.path.isdir(GS), reason="reference tree not mounted")
def test_install_patches_the_real_reference_renderer_module():
    """The REFERENCE's own ``sugar/gaussian_splatting/gaussian_renderer/__init__.py``, imported unchanged from where it lies
    (its heavy imports stubbed as tests/test_render_mirror.py does) in a fresh interpreter after ``autovfx_amd.install()``:
    its ``diff_gaussian_rasterization`` is this repository's, its ``render`` is replaced, the original is kept, and the
    signatures agree (a caller's positional / keyword use keeps working)."""
    code = textwrap.dedent(f"""
        import inspect, sys, types
        sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        import autovfx_amd
        autovfx_amd.install()
        from test_render_mirror import _import_reference
        gr = _import_reference("gaussian_renderer")          # `from gaussian_renderer import render` (train.py:16)
        import diff_gaussian_rasterization as dgr
        ours, theirs = inspect.signature(gr.render), inspect.signature(gr.reference_render)
        print(gr.render.__module__, gr.reference_render.__module__, dgr.__file__.startswith({ROOT!r}),
              list(ours.parameters) == list(theirs.parameters),
              [p.default for p in ours.parameters.values()][4:] == [p.default for p in theirs.parameters.values()][4:])
    """)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.split() == ["autovfx_amd.renderer", "gaussian_renderer", "True", "True", "True"], r.stdout
