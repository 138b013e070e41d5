"""The C oracle against the committed golden vectors (tests/golden/*.npz).

The vectors were produced by the REFERENCE's own sources compiled for the host
(tests/golden/make_golden.py -> oracle/_ref); this is the pin that SURVEY.md section 8c asks for
in the absence of reference-side tests.  Bar: bit-exact, every output and every intermediate.
"""
import glob
import os

import numpy as np
import pytest

from oracle import cpu_oracle

GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
                if not os.path.basename(p).startswith(("bw_", "ply_")))


def load_case(path):
    z = np.load(path)
    kw = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    for k in ("width", "height", "sh_degree"):
        kw[k] = int(kw[k])
    for k in ("tanfovx", "tanfovy", "scale_modifier"):
        kw[k] = float(kw[k])
    ref = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    ref["num_rendered"] = int(ref["num_rendered"])
    return kw, ref


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_bit_identical(got, ref, name, visible_only=("means2D", "depths", "conic_opacity", "rgb")):
    vis = ref["radii"] > 0
    assert got["num_rendered"] == ref["num_rendered"], name
    for k, r in ref.items():
        if k == "num_rendered":
            continue
        g = got[k]
        if k in visible_only:   # rows of culled Gaussians are unspecified scratch
            g, r = g[vis], r[vis]
        assert g.shape == r.shape, f"{name}:{k} shape {g.shape} vs {r.shape}"
        neq = int((bits(g) != bits(r)).sum())
        assert neq == 0, f"{name}:{k} differs in {neq} elements"


def test_golden_files_present():
    assert len(GOLDEN) >= 8, "golden fixtures missing; run tests/golden/make_golden.py in the build container"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_vectors(path):
    kw, ref = load_case(path)
    got = cpu_oracle.forward(intermediates=True, **kw)
    assert_bit_identical(got, ref, os.path.basename(path))
