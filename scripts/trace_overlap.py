"""Overlap analysis of a rocprofv3 kernel trace (multi-stream bench run): union busy time, mean concurrency and how
much each kernel stretches under contention.  usage: python scripts/trace_overlap.py <run_kernel_trace.csv>"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); n = re.sub(r"^void ", "", n).split("(")[0].replace("gsr::", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Stream_Id", r.get("Queue_Id", "?"))))
ev.sort()
# steady-state window: middle 60 % of the run
t_lo = ev[len(ev) // 5][0]; t_hi = ev[len(ev) * 4 // 5][1]
win = [e for e in ev if e[0] >= t_lo and e[1] <= t_hi]
pts = sorted([(s, 1) for s, e, *_ in win] + [(e, -1) for s, e, *_ in win])
busy = 0; conc_time = 0; cur = 0; last = pts[0][0]
for t, d in pts:
    if cur > 0: busy += t - last; conc_time += (t - last) * cur
    cur += d; last = t
span = t_hi - t_lo
print(f"window {span / 1e6:.2f} ms, {len(win)} kernels, GPU busy {100 * busy / span:.1f} %, mean concurrency while busy {conc_time / busy:.2f}")
d = defaultdict(list)
for s, e, n, q in win: d[n].append(e - s)
nblend = max(1, sum(len(v) for n, v in d.items() if n.startswith("preprocess_kernel")))   # frames = projection launches
print(f"frames in window ~{nblend}: {span / nblend / 1e3:.1f} us per frame; sum of kernel durations per frame {sum(sum(v) for v in d.values()) / nblend / 1e3:.1f} us")
for n, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:18]:
    print(f"  {n[:44]:44s} calls/frame {len(v) / nblend:5.2f}  mean {sum(v) / len(v) / 1e3:8.1f} us  total/frame {sum(v) / nblend / 1e3:8.1f} us")
