"""Instruction census of blend_quadrant_kernel's inner loop from the compiler's own assembly (no GPU needed).

Compiles autovfx_amd/csrc/gsr_blend.hip with the library's flags to gfx950 assembly, finds the innermost loop of
blend_quadrant_kernel<false> (the `while (todo)` walk over the staged entries that reach the quadrant) and splits it
into the stages an iteration can stop at:
    A  every processed entry : record reads, dx / dy, power, the two range tests
    B  some pixel in range   : exp, alpha, the 1/255 test
    C  some pixel blends     : T (1 - alpha), the 1e-4 test
    D  some pixel accumulates: colour read, the four packed multiply-adds, T and last-contributor update
and counts per stage VALU (of which comparisons / transcendental / packed), SALU, LDS and wait instructions.
    python scripts/blend_isa_census.py [--markdown]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autovfx_amd import build  # noqa: E402


def assembly():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "blend.s")
        flags = [f for f in build.FLAGS if f not in ("-fPIC",)]
        subprocess.check_call([build.hipcc(), "-x", "hip", *flags, "--cuda-device-only", "-S", "-o", out,
                               os.path.join(build.CSRC, "gsr_blend.hip")], stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_body(asm, mangled_part="blend_quadrant_kernelILb0"):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and mangled_part in l and l.rstrip().endswith(":") or
                 (l.startswith("_Z") and mangled_part in l and ": " in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".amdhsa_kernel") or lines[i].strip() == ".end_amdhsa_kernel" or
               lines[i].strip().startswith(".section") and i > start + 5)
    return lines[start:end]


def classify(op):
    if op.startswith(("v_cmp", "v_cmpx")):
        return "v_cmp"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def main():
    body = kernel_body(assembly())
    # innermost loop: the block whose header comment says "Inner Loop Header: Depth=2"
    hdr = next(i for i, l in enumerate(body) if "Inner Loop Header: Depth=2" in l)
    label = None
    for j in range(hdr, -1, -1):
        m = re.match(r"^(\.LBB\d+_\d+):", body[j])
        if m:
            label = m.group(1)
            break
    # the loop runs from its header label to the backward branch to it
    first = next(i for i, l in enumerate(body) if l.startswith(label + ":"))
    last = max(i for i, l in enumerate(body) if re.search(r"s_cbranch_\w+\s+" + re.escape(label) + r"\b", l) and i > first)
    loop = body[first:last + 1]
    # stages: cut at the conditional branches that leave an iteration early (forward branches inside the loop)
    stages, cur = [], []
    for l in loop:
        s = l.strip()
        if not s or s.startswith((";", ".LBB", "//")) or s.startswith(";;#"):
            if s.startswith(".LBB"):
                cur.append(("label", s))
            continue
        op = s.split()[0]
        cur.append((classify(op), s))
        if op.startswith("s_cbranch") and len(stages) < 3 and any(c == "v_cmp" for c, _ in cur):
            stages.append(cur)
            cur = []
    stages.append(cur)
    names = ["A every processed entry", "B some pixel in range", "C some pixel blends", "D accumulate + loop tail"]
    kinds = ["valu", "v_cmp", "trans", "v_pk", "salu", "branch", "lds", "wait", "vmem", "other"]
    md = "--markdown" in sys.argv
    rows = []
    for name, st in zip(names, stages):
        cnt = {k: sum(1 for c, _ in st if c == k) for k in kinds}
        cnt["VALU total"] = cnt["valu"] + cnt["v_cmp"] + cnt["trans"] + cnt["v_pk"]
        cnt["SALU total"] = cnt["salu"] + cnt["branch"]
        rows.append((name, cnt))
    cols = ["VALU total", "valu", "v_cmp", "trans", "v_pk", "SALU total", "lds", "wait", "vmem"]
    if md:
        print("| stage | " + " | ".join(cols) + " |")
        print("|---|" + "---|" * len(cols))
        for name, cnt in rows:
            print(f"| {name} | " + " | ".join(str(cnt[c]) for c in cols) + " |")
        tot = {c: sum(cnt[c] for _, cnt in rows) for c in cols}
        print("| **full path** | " + " | ".join(f"**{tot[c]}**" for c in cols) + " |")
    else:
        for name, cnt in rows:
            print(f"{name:28s} " + "  ".join(f"{c}={cnt[c]}" for c in cols))
    if "--dump" in sys.argv:
        print("\n".join(loop))


if __name__ == "__main__":
    main()
