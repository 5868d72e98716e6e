#!/usr/bin/env python3
"""Launch sequence of ONE steady-state eager forward from a rocprofv3 --kernel-trace CSV of tools/forward_one.py: the launches
between the last two `head_conv_*` launches (start offset, gap, duration, workgroups, LDS, kernel) + a per-kernel summary.
usage: prof_seq2.py <dir> [marker]"""
import csv
import glob
import sys
from collections import OrderedDict

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "head_conv_"
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gi = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = gi[-2], gi[-1]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
tot = OrderedDict()
busy = 0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("sdfmm::", "")
    wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 0
    grid = int(r["Grid_Size_X"]) // max(wg, 1) if "Grid_Size_X" in r else 0
    print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:5.1f}  dur {(e - s) / 1e3:7.1f} us  wgs {grid:6d} x {wg:4d}  lds {r.get('LDS_Block_Size', '?'):>6}  {n[:90]}")
    prev_end = max(prev_end, e)
    k = n.split("(")[0][:80]
    a = tot.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    busy += (e - s) / 1e3
period = (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3
print(f"\none forward: period {period:.1f} us, kernel busy {busy:.1f} us, launches {hi - lo}")
for k, (c, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{t:9.1f} us  x {c:3d}  avg {t / c:7.1f} us  {k}")
