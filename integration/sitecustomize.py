"""Opt-in hook for an UNCHANGED AutoVFX checkout: with ``<repo>/integration`` and ``<repo>`` on ``PYTHONPATH`` and
``AUTOVFX_AMD_INSTALL=1`` in the environment, every Python process calls ``autovfx_amd.install()`` at start-up (see
autovfx_amd/hook.py and INTEGRATION.md section 1).  Without the variable this file does nothing.  Python imports only
the first ``sitecustomize`` on ``sys.path``; a site-wide one that this file shadows is chained to below.
"""
import os
import sys


def _chain():
    here = os.path.dirname(os.path.abspath(__file__))
    import importlib.machinery
    rest = [p for p in sys.path if os.path.abspath(p or os.getcwd()) != here]
    spec = importlib.machinery.PathFinder.find_spec("sitecustomize", rest)
    if spec is not None and spec.loader is not None and os.path.abspath(spec.origin or "") != os.path.abspath(__file__):
        import importlib.util
        module = importlib.util.module_from_spec(spec)
        try:
            spec.loader.exec_module(module)
        except Exception:   # a broken site-wide hook must not take the interpreter down, exactly as site.py treats it
            pass


_chain()
if os.environ.get("AUTOVFX_AMD_INSTALL", "") == "1":
    try:
        import autovfx_amd.hook as _install
        _install.install(strict=False)   # a render path that cannot be loaded later costs one stderr line, not the process
    except Exception as e:   # never break interpreter start-up; the process then runs the reference's own path
        sys.stderr.write(f"[autovfx_amd] AUTOVFX_AMD_INSTALL=1 but install() failed: {e!r}\n")
