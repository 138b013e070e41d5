"""Divide rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB) by the byte counts scripts/ubench/pmc_calib.hip knows it moved.
usage: python scripts/pmc_calib_reduce.py <dir with fetch/ write/ passes and table.csv> <out.csv>"""
import csv, glob, os, re, sys

src, dst = sys.argv[1], sys.argv[2]
known = {}
with open(os.path.join(src, "table.csv")) as f:
    for r in csv.DictReader(l for l in f if "," in l and "amdgpu" not in l):
        known[r["kernel"]] = (int(r["asked_bytes"]), int(r["line64_bytes"]))
vals = {}
for path in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
            if name in known:
                vals.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(dst, "w") as f:
    f.write("# scripts/ubench/pmc_calib.hip under rocprofv3 --pmc (one counter per pass), 1 GiB tables, MI355X\n")
    f.write("# counter_bytes = counter * its unit (KiB counters * 1024, n-byte request counters * n); ratio_asked = counter_bytes / bytes the lanes asked for; ratio_lines = / bytes of the 64-byte lines touched\n")
    f.write("kernel,counter,counter_bytes,asked_bytes,line64_bytes,ratio_asked,ratio_lines\n")
    unit = {"FETCH_SIZE": 1024.0, "WRITE_SIZE": 1024.0, "TCC_EA0_RDREQ_32B_sum": 32.0, "TCC_EA0_RDREQ_64B_sum": 64.0,
            "TCC_EA0_RDREQ_128B_sum": 128.0, "TCC_EA0_RDREQ_DRAM_32B_sum": 32.0, "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum": 32.0,
            "TCC_BUBBLE_sum": 128.0, "TCC_EA0_WRREQ_64B_sum": 64.0, "TCC_EA0_RDREQ_sum": 0.0, "TCC_EA0_WRREQ_sum": 0.0}
    for (k, c), v in sorted(vals.items()):
        if unit.get(c, 0.0) == 0.0:   # plain request counts: listed as counts
            f.write(f"{k},{c} (requests),{sum(v) / len(v):.0f},{known[k][0]},{known[k][1]},,\n")
            continue
        b = sum(v) / len(v) * unit[c]
        asked, lines = known[k]
        f.write(f"{k},{c},{b:.0f},{asked},{lines},{b / asked:.3f},{b / lines:.3f}\n")
print(open(dst).read())
