"""Reduce the rocprofv3 --pmc passes written by scripts/gpu_pmc.sh to one small CSV: mean counter value per launch
for each of this library's kernels.  usage: python scripts/pmc_reduce.py gpurun_out/pmc profiles/r01_pmc_per_kernel_mean.csv"""
import csv, glob, os, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(src, "*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if "gsr::" not in name:
                continue
            short = re.sub(r"\(anonymous namespace\)::", "", name)
            short = re.sub(r"^void ", "", short)
            short = short.split("(")[0].replace("gsr::", "")
            a = acc[(short, r["Counter_Name"])]
            a[0] += float(r["Counter_Value"]); a[1] += 1
# a kernel dispatch is reported once per XCD / dimension by some counters: normalise by launches of the kernel
with open(dst, "w") as f:
    f.write("# rocprofv3 --pmc passes (one counter group per run, kernel-trace only), bench.py --steps 6 --warmup 2 --streams 1, C3 workload\n")
    f.write("# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM)\n")
    f.write("kernel,counter,mean_per_launch,launches\n")
    for (k, c), (tot, n) in sorted(acc.items()):
        f.write(f"{k},{c},{tot / n:.1f},{n}\n")
print(open(dst).read()[:3000])
