"""The drop-in Python surface equals the reference's.

The reference package cannot be imported here (its ``_C`` is a CUDA extension), so its source is
parsed with ``ast`` when /root/reference is mounted, and a committed snapshot of the same facts
(tests/golden/api_surface.json, written by this file when run as a script) is used elsewhere.
"""
import ast
import inspect
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_INIT = ("/root/reference/sugar/gaussian_splatting/submodules/diff-gaussian-rasterization/"
            "diff_gaussian_rasterization/__init__.py")
SNAPSHOT = os.path.join(ROOT, "tests", "golden", "api_surface.json")


def surface_from_source(src: str) -> dict:
    tree = ast.parse(src)
    out = {"functions": {}, "classes": {}}
    args_of = lambda fn: [a.arg for a in fn.args.args] + ([f"*{fn.args.vararg.arg}"] if fn.args.vararg else [])
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and not node.name.startswith("cpu_deep"):
            out["functions"][node.name] = args_of(node)
        elif isinstance(node, ast.ClassDef):
            cls = {"bases": [ast.unparse(b) for b in node.bases], "methods": {}, "fields": []}
            for item in node.body:
                if isinstance(item, ast.FunctionDef):
                    cls["methods"][item.name] = args_of(item)
                elif isinstance(item, ast.AnnAssign):
                    cls["fields"].append([item.target.id, ast.unparse(item.annotation)])
            out["classes"][node.name] = cls
    return out


def ours() -> dict:
    with open(os.path.join(ROOT, "diff_gaussian_rasterization", "__init__.py")) as f:
        return surface_from_source(f.read())


def check(ref: dict):
    mine = ours()
    assert mine["functions"]["rasterize_gaussians"] == ref["functions"]["rasterize_gaussians"]
    for cname, rc in ref["classes"].items():
        mc = mine["classes"][cname]
        assert mc["bases"] == rc["bases"], cname
        assert mc["fields"] == rc["fields"], cname
        for mname, margs in rc["methods"].items():
            if cname.startswith("_"):   # private autograd node: same arity, argument names are not API
                assert len(mc["methods"][mname]) == len(margs), f"{cname}.{mname}"
            else:
                assert mc["methods"][mname] == margs, f"{cname}.{mname}"


def test_surface_matches_committed_snapshot():
    with open(SNAPSHOT) as f:
        check(json.load(f))


@pytest.mark.skipif(not os.path.exists(REF_INIT), reason="reference tree not mounted")
def test_surface_matches_reference_source_and_snapshot_is_fresh():
    with open(REF_INIT) as f:
        ref = surface_from_source(f.read())
    check(ref)
    with open(SNAPSHOT) as f:
        assert json.load(f) == ref, "tests/golden/api_surface.json is stale; run python tests/test_api_compat.py"


def test_native_module_surface():
    from diff_gaussian_rasterization import _C, GaussianRasterizationSettings
    # DGR/ext.cpp:16-18 and DGR/rasterize_points.h:18-38,67-70
    # the reference's 19 positional arguments, in its order; one keyword-only addition that defaults to its behaviour
    params = inspect.signature(_C.rasterize_gaussians).parameters
    positional = [n for n, p in params.items() if p.kind is not inspect.Parameter.KEYWORD_ONLY]
    assert positional == [
        "background", "means3D", "colors", "opacity", "scales", "rotations", "scale_modifier", "cov3D_precomp",
        "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "image_height", "image_width", "sh", "degree", "campos",
        "prefiltered", "debug"]
    assert [n for n in params if n not in positional] == ["inference"] and params["inference"].default is False
    assert list(inspect.signature(_C.mark_visible).parameters) == ["means3D", "viewmatrix", "projmatrix"]
    bw_params = inspect.signature(_C.rasterize_gaussians_backward).parameters
    # (one keyword-only addition that defaults to the reference's behaviour: every gradient tensor is written and returned)
    assert [n for n, p in bw_params.items() if p.kind is inspect.Parameter.KEYWORD_ONLY] == ["skip_unused"]
    assert bw_params["skip_unused"].default is False
    assert [n for n, p in bw_params.items() if p.kind is not inspect.Parameter.KEYWORD_ONLY] == [
        "background", "means3D", "radii", "colors", "scales", "rotations", "scale_modifier", "cov3D_precomp",
        "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "dL_dout_color", "dL_dout_depth", "dL_dout_alpha", "sh",
        "degree", "campos", "geomBuffer", "R", "binningBuffer", "imageBuffer", "out_alpha", "debug"]
    assert GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


if __name__ == "__main__":
    with open(REF_INIT) as f:
        snap = surface_from_source(f.read())
    with open(SNAPSHOT, "w") as f:
        json.dump(snap, f, indent=1, sort_keys=True)
    print("wrote", SNAPSHOT)
