#!/usr/bin/env python3
"""How the three in-flight forwards share the chip: from a rocprofv3 --kernel-trace of the default bench command, over the timed
region's launches - per kernel name the total / average duration UNDER OVERLAP, and a sweep of the timeline: the share of wall
time in which 0, 1, 2, 3+ kernels run, and in which at least one 'whole-chip' kernel (more than 96 KiB of LDS per workgroup or
more than 600 threads: it fills a compute unit's registers / LDS and admits nothing beside it) runs.  usage: overlap_report.py <dir>"""
import csv, glob, sys, collections, re
d = sys.argv[1]
rows = [r for f in glob.glob(d + "/**/*_kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the last 40 % of the trace's graph-replayed launches (the bench's side measurements come after it, so cut by name density)
heads = [i for i, r in enumerate(rows) if "head_conv_sn_kernel" in r["Kernel_Name"]]
lo, hi = heads[len(heads) // 2], heads[len(heads) // 2 + 60] if len(heads) // 2 + 60 < len(heads) else heads[-1]
sel = rows[lo:hi]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
nfw = sum("head_conv_sn_kernel" in r["Kernel_Name"] for r in sel)
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("sdfmm::", "")
    return re.sub(r"\(.*", "", n)[:60]
def whole(r):
    return int(r.get("LDS_Block_Size", 0) or 0) > 96 * 1024 or int(r.get("Workgroup_Size_X", 0) or 0) > 600
ev = []
per = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1, whole(r))); ev.append((e, -1, whole(r)))
    k = short(r["Kernel_Name"]) + (" *" if whole(r) else "")
    per[k][0] += 1; per[k][1] += (e - s) / 1e3
ev.sort()
cur = curw = 0
last = t0
share = collections.Counter(); wshare = collections.Counter()
for t, dlt, w in ev:
    share[min(cur, 3)] += t - last; wshare[min(curw, 2)] += t - last
    last = t
    cur += dlt; curw += dlt if w else 0
tot = t1 - t0
print(f"{nfw} forwards in {tot / 1e6:.2f} ms = {tot / 1e3 / nfw:.1f} us per forward under overlap; sum of kernel durations {sum(v[1] for v in per.values()) / nfw:.0f} us per forward")
print("kernels running at once: " + ", ".join(f"{k}{'+' if k == 3 else ''}: {100 * v / tot:.1f} %" for k, v in sorted(share.items())))
print("whole-chip kernels (*) running at once: " + ", ".join(f"{k}{'+' if k == 2 else ''}: {100 * v / tot:.1f} %" for k, v in sorted(wshare.items())))
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{t / nfw:8.1f} us/forward  x {c / nfw:5.1f}  avg {t / c:7.1f} us  {k}")
