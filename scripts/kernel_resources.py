#!/usr/bin/env python
"""Registers, LDS and the occupancy they allow for EVERY kernel of the shipped library, read from the code object itself.

    python scripts/kernel_resources.py [autovfx_amd/lib/libgsr_hip.so] > profiles/r06_kernel_resources.csv

The .so carries its device code in the ``.hip_fatbin`` section as a clang offload bundle; every ``hipv4-amdgcn-...gfx950`` entry
is an ELF whose ``NT_AMDGPU_METADATA`` note (``llvm-readelf --notes``) lists, per kernel, ``.vgpr_count``, ``.agpr_count``,
``.sgpr_count``, ``.group_segment_fixed_size`` (static LDS), ``.private_segment_fixed_size`` (scratch), ``.vgpr_spill_count``.
Occupancy follows /opt/skills/guides/MI355X_MICROARCH.md ("Register files"): the allocation granule is 8 registers per lane,
waves per SIMD by registers = min(8, floor(512 / alloc)); workgroups per CU by LDS = floor(160 KiB / LDS per workgroup); 32 waves
per CU.  (Round 5's notes quoted 129 VGPRs for ``preprocess_backward_kernel``: that was the profiler's ``VGPR_Count`` column of
another build -- ``arch_vgpr_count + accum_vgpr_count`` rounded -- not this binary; the table below is the binary.)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
CXXFILT = "c++filt"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(path):
    """[(triple, bytes)] of every device code object bundled in the file."""
    data = open(path, "rb").read()
    out, at = [], 0
    while True:
        at = data.find(MAGIC, at)
        if at < 0:
            return out
        n, = struct.unpack_from("<Q", data, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if size and "amdgcn" in triple:
                out.append((triple, data[at + off:at + off + size]))
        at += len(MAGIC)


def kernels_of(elf_bytes):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(elf_bytes)
        f.flush()
        notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    rows, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2).strip().strip("'")
        if key == "agpr_count" or (key == "args" and cur is None):
            pass
        if re.match(r"\s*- \.", line) and key in ("agpr_count", "args"):    # first key of a kernel's map (keys are sorted)
            cur = {}
            rows.append(cur)
        if cur is not None and key in ("agpr_count", "vgpr_count", "sgpr_count", "group_segment_fixed_size", "private_segment_fixed_size",
                                       "max_flat_workgroup_size", "name", "vgpr_spill_count", "sgpr_spill_count", "wavefront_size",
                                       "kernarg_segment_size"):
            cur[key] = val
    return [r for r in rows if "name" in r and "vgpr_count" in r]


def wps_full(wgs, waves_per_wg):
    return wgs * waves_per_wg >= 32


def occupancy(vgpr, agpr, lds, wg_threads, sgpr=0):
    alloc = max(8, -(-(vgpr + agpr) // 8) * 8)
    by_regs = min(8, 512 // alloc)
    waves_per_wg = max(1, -(-wg_threads // 64))
    wgs_by_regs = (by_regs * 4) // waves_per_wg
    wgs_by_lds = (160 * 1024) // lds if lds else 10 ** 6
    wgs_by_waves = 32 // waves_per_wg
    if sgpr and wg_threads == 256:     # guide, "Residency": 256-thread blocks per CU <= floor(800 / (ceil(sgpr / 16) * 16 + 16))
        wgs_by_waves = min(wgs_by_waves, 800 // (-(-sgpr // 16) * 16 + 16))
    wgs = max(0, min(wgs_by_regs, wgs_by_lds, wgs_by_waves))
    limiter = "+".join(n for n, v in (("vgpr", wgs_by_regs), ("lds", wgs_by_lds), ("waves/sgpr", wgs_by_waves)) if v == wgs)
    if wps_full(wgs, waves_per_wg):
        limiter = "none (8 waves/SIMD)"
    return alloc, by_regs, wgs, wgs * waves_per_wg / 4.0, limiter


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "autovfx_amd", "lib", "libgsr_hip.so")
    print("kernel,vgpr_count,agpr_count,vgpr_alloc,sgpr_count,lds_bytes,scratch_bytes,vgpr_spills,max_workgroup,waves_per_simd_by_vgpr,"
          "workgroups_per_cu,waves_per_simd,limited_by")
    seen = set()
    for triple, blob in code_objects(lib):
        if "gfx950" not in triple:
            continue
        for k in kernels_of(blob):
            name = k["name"]
            try:
                name = subprocess.run([CXXFILT, name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            name = name.replace("gsr::(anonymous namespace)::", "").replace("gsr::", "").replace("void ", "")
            name = name[:name.index("(")] if "(" in name else name
            if name in seen:
                continue
            seen.add(name)
            v, a, lds = int(k["vgpr_count"]), int(k.get("agpr_count", 0)), int(k.get("group_segment_fixed_size", 0))
            wg = int(k.get("max_flat_workgroup_size", 256))
            alloc, by_regs, wgs, wps, lim = occupancy(v, a, lds, wg, int(k.get('sgpr_count', 0) or 0))
            print(",".join(str(x) for x in ('"' + name + '"', v, a, alloc, k.get("sgpr_count", ""), lds, k.get("private_segment_fixed_size", 0),
                                            k.get("vgpr_spill_count", 0), wg, by_regs, wgs, wps, lim)))


if __name__ == "__main__":
    main()
