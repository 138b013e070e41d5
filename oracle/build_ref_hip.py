"""Compile the REFERENCE's own CUDA rasterizer for gfx950 with hipcc -> oracle/_ref/libgsr_ref_hip.so.

TEST / BENCH INFRASTRUCTURE ONLY -- a measuring stick, never part of the product (the product is hand-written
HIP; nothing here is hipified into it).  It answers two questions on the same MI355X:
  * do our kernels agree with the reference's kernels executed on this GPU (tests/test_reference_hip_gpu.py),
  * how fast is the reference's own pipeline here (bench.py --reference-hip), i.e. what "matching the reference"
    means in frames per second.
How: the reference's three .cu files are compiled from where they lie under /root/reference.  They use a dozen CUDA
runtime names and two CUB calls; oracle/ref_hip_shim/ maps those onto HIP / hipCUB with macros and one namespace
alias.  The only thing clang cannot parse is the spaced launch syntax ``kernel << <grid, block >> > (args)``; as in
build_ref.py those expressions are rewritten in memory (to hipLaunchKernelGGL).  hipcc compiles a translation unit
twice (host and device), so it cannot read it from a pipe: the rewritten text goes to a temporary directory outside
the repository that is deleted when the build ends.  No reference source is copied into the repository; only the .so
lands in oracle/_ref/ (git-ignored; it travels to the GPU box, where /root/reference does not exist).
Compiled with -ffp-contract=off: the parity contract of this repository is "the reference's source, IEEE fp32,
unfused" (DESIGN.md section 2), and with it the integer outputs (radii, pair counts) can be compared exactly.
"""
from __future__ import annotations

import os
import subprocess
import sys
import tempfile

from .build_ref import REFERENCE_DGR, CUDA_SOURCES, _LAUNCH, _split_top_level  # noqa: F401  (same sources, same parser)
import re

HERE = os.path.dirname(os.path.abspath(__file__))
SHIM = os.path.join(HERE, "ref_hip_shim")
OUT_DIR = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT_DIR, "libgsr_ref_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-ffp-contract=off"]


def rewrite_launches_hip(src: str) -> str:
    """kernel<T> << <G, B >> > (args)  ->  hipLaunchKernelGGL((kernel<T>), dim3(G), dim3(B), 0, 0, args)"""
    out, pos = [], 0
    while True:
        m = _LAUNCH.search(src, pos)
        if not m:
            out.append(src[pos:])
            return "".join(out)
        kernel = m.group(1)
        cfg_end = re.compile(r">>\s*>").search(src, m.end())
        cfg = _split_top_level(src[m.end():cfg_end.start()])
        if len(cfg) != 2:
            raise ValueError(f"unsupported launch configuration: {cfg}")
        i = cfg_end.end()
        while src[i].isspace():
            i += 1
        depth, j = 0, i
        while True:
            depth += src[j] == "("
            depth -= src[j] == ")"
            if depth == 0:
                break
            j += 1
        out.append(src[pos:m.start()])
        out.append(f"hipLaunchKernelGGL(({kernel}), dim3({cfg[0]}), dim3({cfg[1]}), 0, 0, {src[i + 1:j]})")
        pos = j + 1


def build(force: bool = False, verbose: bool = False) -> str:
    src_dir = os.path.join(REFERENCE_DGR, "cuda_rasterizer")
    if not os.path.isdir(src_dir):
        raise FileNotFoundError(f"{src_dir}: the reference tree is not mounted (it never is on the GPU box)")
    api = os.path.join(SHIM, "ref_hip_api.cpp")
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
                tu = rewrite_launches_hip(f.read())
            obj, tmp_src = os.path.join(tmp, name + ".o"), os.path.join(tmp, name + ".hip")
            with open(tmp_src, "w") as f:
                f.write(tu)
            cmd = [HIPCC, *FLAGS, *inc, "-x", "hip", "-c", tmp_src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=src_dir)
            os.remove(tmp_src)
            objs.append(obj)
        obj = os.path.join(tmp, "ref_hip_api.o")
        subprocess.run([HIPCC, *FLAGS, *inc, "-x", "hip", "-c", api, "-o", obj], check=True)
        objs.append(obj)
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
