"""Reduce the rocprofv3 --pmc passes written by scripts/gpu_pmc.sh to one small CSV: mean counter value per launch
for each of this library's kernels.  usage: python scripts/pmc_reduce.py gpurun_out/pmc profiles/r01_pmc_per_kernel_mean.csv"""
import csv, glob, os, re, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
# the three radix kernels serve both sorts of a frame: a launch over ceil(P / 4096) tiles belongs to the depth sort
P_GAUSSIANS = int(os.environ.get("GSR_PMC_P", "3000000"))
DEPTH_GRID = ((P_GAUSSIANS + 4095) // 4096) * 256
by_sort = defaultdict(lambda: [0.0, 0])   # (kernel, counter, "depth" | "tile") -> [sum, launches]
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
            if short.startswith(("radix_count_kernel", "radix_scatter_kernel")):
                b = by_sort[(short.split("<")[0], r["Counter_Name"], "depth" if int(r["Grid_Size"]) == DEPTH_GRID else "tile")]
                b[0] += float(r["Counter_Value"]); b[1] += 1
# a kernel dispatch is reported once per XCD / dimension by some counters: normalise by launches of the kernel
with open(dst, "w") as f:
    f.write("# rocprofv3 --pmc passes (one counter group per run, kernel-trace only), bench.py --steps 6 --warmup 2 --streams 1, C3 workload\n")
    f.write("# FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-counts wide coalesced reads by 2x (MI355X_MICROARCH.md, HBM)\n")
    f.write("kernel,counter,mean_per_launch,launches\n")
    for (k, c), (tot, n) in sorted(acc.items()):
        f.write(f"{k},{c},{tot / n:.1f},{n}\n")
print(open(dst).read()[:3000])

# ---- HBM traffic per launch and per frame (bench.py's roofline.traffic / roofline.frame.traffic) ----
# usage: ... pmc_reduce.py <passes dir> <out.csv> [<traffic.json> <commit>]
if len(sys.argv) >= 5:
    import json
    fetch = {k: v for (k, c), v in acc.items() if c == "FETCH_SIZE"}
    write = {k: v for (k, c), v in acc.items() if c == "WRITE_SIZE"}
    frames = next((n for k, (tot, n) in fetch.items() if k.startswith("preprocess_kernel")), 0)
    per_variant, frame_bytes = {}, 0.0
    for k in sorted(set(fetch) & set(write)):
        f_kib, nf = fetch[k][0] / fetch[k][1], fetch[k][1]
        w_kib = write[k][0] / write[k][1]
        b = (2.0 * f_kib + w_kib) * 1024.0   # every L2 miss is a 128-byte request tallied at 64 (profiles/r02_pmc_calibration.csv)
        per_variant[k] = {"bytes_per_launch": b, "launches_per_frame": nf / frames if frames else None,
                          "fetch_kib": f_kib, "write_kib": w_kib}
        if frames:
            frame_bytes += b * nf / frames
    kernels = {}
    for k, v in per_variant.items():   # template variants of one kernel: launch-weighted mean
        base = k.split("<")[0]
        e = kernels.setdefault(base, {"bytes": 0.0, "launches": 0.0})
        e["bytes"] += v["bytes_per_launch"] * (v["launches_per_frame"] or 0.0)
        e["launches"] += v["launches_per_frame"] or 0.0
    kernels = {k: {"bytes_per_launch": e["bytes"] / e["launches"] if e["launches"] else 0.0, "launches_per_frame": e["launches"]}
               for k, e in kernels.items()}
    # HBM-side bytes per frame and pipeline stage (bench.py's per-stage counter figures)
    stage_of = {"preprocess_kernel": "preprocess", "counter_tally_kernel": "preprocess", "bin_gather_kernel": "duplicate", "bin_offsets_kernel": "duplicate",
                "slab_recount_kernel": "duplicate", "slab_compact_kernel": "duplicate", "expand_kernel": "duplicate",
                "tile_ranges_kernel": "ranges", "sh_colour_listed_kernel": "colour", "sh_colour_all_kernel": "colour",
                "blend_quadrant_kernel": "blend"}
    stages = defaultdict(float)
    for k, e in kernels.items():
        if k in stage_of:
            stages[stage_of[k]] += e["bytes_per_launch"] * e["launches_per_frame"]
    sort_launches = {"depth": 0.0, "tile": 0.0}
    for kern in ("radix_count_kernel", "radix_scatter_kernel"):
        for which in ("depth", "tile"):
            fs, ws = by_sort.get((kern, "FETCH_SIZE", which)), by_sort.get((kern, "WRITE_SIZE", which))
            if fs and ws and frames:
                stages[which + "_sort"] += (2.0 * fs[0] + ws[0]) * 1024.0 / frames
                if kern == "radix_count_kernel":
                    sort_launches[which] = fs[1] / frames
    scan = kernels.get("radix_scan_kernel")
    if scan and sum(sort_launches.values()) > 0:   # the scan kernel's grid does not depend on the item count: split by launches
        tot = scan["bytes_per_launch"] * scan["launches_per_frame"]
        for which in ("depth", "tile"):
            stages[which + "_sort"] += tot * sort_launches[which] / sum(sort_launches.values())
    with open(sys.argv[3], "w") as f:
        json.dump({"commit": sys.argv[4], "workload": "c3", "command": "bench.py --profile-run --steps 6 --warmup 2 --streams 1",
                   "formula": "(2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024, mean per launch; frame = sum over kernels x launches per frame",
                   "frames_profiled": frames, "frame_bytes": frame_bytes, "stages": dict(stages), "kernels": kernels,
                   "variants": per_variant}, f, indent=1)
    print("frame traffic bytes:", int(frame_bytes))
