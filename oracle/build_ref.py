"""Compile the REFERENCE's own rasterizer sources for the host -> oracle/_ref/libgsr_ref.so.

TEST INFRASTRUCTURE ONLY.  The reference path is CUDA, and this image has neither nvcc nor an
NVIDIA device, but the path's sources are few and self-contained (GLM is vendored; CUB's two calls
have published contracts), so they can be compiled by g++ from where they lie under
/root/reference once four things are supplied by oracle/ref_shim/:
  * the CUDA headers they include (qualifiers as empty macros, vector types, min/max overloads),
  * cooperative_groups' this_grid()/this_thread_block(),
  * cub::DeviceScan::InclusiveSum / cub::DeviceRadixSort::SortPairs (restated contracts),
  * a fiber runtime that executes a grid one CUDA thread at a time with real barriers.
The only thing g++ cannot parse is the ``kernel<<<grid, block>>>(args)`` launch syntax; this script
rewrites those expressions IN MEMORY to ``gsr_shim::launch(grid, block, [&]{ kernel(args); })`` and
pipes the translation unit to g++ on stdin.  No reference source is copied into the repository or
written to disk; only the .so lands in oracle/_ref/ (git-ignored).

Compiled with -ffp-contract=off: nvcc would contract a*b+c into FMAs in places nobody can observe
here, so the pin is "the reference's source, IEEE fp32, unfused" -- the same contract the C oracle
and the HIP kernels are built to.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "ref_shim")
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libgsr_ref.so")
REFERENCE_DGR = os.environ.get(
    "GSR_REFERENCE_DGR",
    "/root/reference/sugar/gaussian_splatting/submodules/diff-gaussian-rasterization")
CUDA_SOURCES = ("forward.cu", "backward.cu", "rasterizer_impl.cu")
CXXFLAGS = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-w"]

_LAUNCH = re.compile(r"([A-Za-z_]\w*(?:\s*<\s*\w+\s*>)?)\s*<<\s*<")


def _split_top_level(text: str):
    parts, depth, cur = [], 0, []
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
    parts.append("".join(cur))
    return [p.strip() for p in parts]


def rewrite_launches(src: str) -> str:
    """kernel<T> << <G, B >> > (args)  ->  gsr_shim::launch(G, B, [&]{ kernel<T>(args); })"""
    out, pos = [], 0
    while True:
        m = _LAUNCH.search(src, pos)
        if not m:
            out.append(src[pos:])
            return "".join(out)
        kernel = m.group(1)
        cfg_start = m.end()
        cfg_end = re.compile(r">>\s*>").search(src, cfg_start)
        if cfg_end is None:
            raise ValueError("unterminated launch configuration")
        cfg = _split_top_level(src[cfg_start:cfg_end.start()])
        if len(cfg) != 2:
            raise ValueError(f"unsupported launch configuration: {cfg}")
        i = cfg_end.end()
        while src[i].isspace():
            i += 1
        if src[i] != "(":
            raise ValueError("launch without argument list")
        depth, j = 0, i
        while True:
            if src[j] == "(":
                depth += 1
            elif src[j] == ")":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        args = src[i + 1:j]
        out.append(src[pos:m.start()])
        out.append(f"gsr_shim::launch(dim3({cfg[0]}), dim3({cfg[1]}), [&]() {{ {kernel}({args}); }})")
        pos = j + 1


def build(force: bool = False, verbose: bool = False) -> str:
    src_dir = os.path.join(REFERENCE_DGR, "cuda_rasterizer")
    if not os.path.isdir(src_dir):
        raise FileNotFoundError(f"{src_dir}: the reference tree is not mounted (it never is on the GPU box)")
    deps = [os.path.join(src_dir, f) for f in os.listdir(src_dir)] + \
           [os.path.join(dp, f) for dp, _, fs in os.walk(SHIM) for f in fs] + [__file__]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    inc = ["-I", SHIM, "-I", src_dir, "-I", os.path.join(REFERENCE_DGR, "third_party", "glm")]
    with tempfile.TemporaryDirectory() as tmp:
        objs = []
        for name in CUDA_SOURCES:
            with open(os.path.join(src_dir, name), "r") as f:
                tu = rewrite_launches(f.read())
            obj = os.path.join(tmp, name + ".o")
            cmd = ["g++", *CXXFLAGS, *inc, "-x", "c++", "-c", "-", "-o", obj]
            if verbose:
                print(" ".join(cmd), "<", name, flush=True)
            subprocess.run(cmd, input=tu.encode(), check=True, cwd=src_dir)
            objs.append(obj)
        for name in ("shim_runtime.cpp", "ref_api.cpp"):
            obj = os.path.join(tmp, name + ".o")
            subprocess.run(["g++", *CXXFLAGS, *inc, "-c", os.path.join(SHIM, name), "-o", obj], check=True)
            objs.append(obj)
        subprocess.run(["g++", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
