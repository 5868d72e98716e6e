#!/usr/bin/env python3
"""Print the launch sequence of ONE steady-state forward from a rocprofv3 --kernel-trace CSV of bench.py:
start offset, duration, grid, LDS, kernel.   usage: prof_sequence.py <dir> [marker] [launches-per-forward]"""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "qk_gate_kernel"
per_fwd = int(sys.argv[3]) if len(sys.argv) > 3 else 12
gi = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
nf = len(gi) // per_fwd
lo, hi = gi[per_fwd * (nf - 2)], gi[per_fwd * (nf - 1)]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    wg = int(r["Workgroup_Size_X"]) if "Workgroup_Size_X" in r else 0
    grid = int(r["Grid_Size_X"]) // max(wg, 1) if "Grid_Size_X" in r else 0
    print(f"{(s - t0) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:5.1f}  dur {(e - s) / 1e3:7.1f} us  wgs {grid:6d}  lds {r.get('LDS_Block_Size', '?'):>6}  {n}")
    prev_end = e
