"""``plyfile`` as tests/ref_plyfile_stub.py stands in for it (binary little-endian vertex tables)."""
from ref_plyfile_stub import PlyData, PlyElement, PlyProperty  # noqa: F401
