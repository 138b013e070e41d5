"""MI355X-native 3D-Gaussian-splatting render path behind the reference's rasterizer API.

Importing the package loads nothing heavy; ``autovfx_amd.install()`` is the one-call integration for an unchanged
AutoVFX process (autovfx_amd/hook.py).
"""


def install(path: bool = True) -> None:
    """Route ``diff_gaussian_rasterization`` and every ``...gaussian_renderer.render`` of this process to this package."""
    from .hook import install as _install
    _install(path)


def uninstall() -> None:
    from .hook import uninstall as _uninstall
    _uninstall()
