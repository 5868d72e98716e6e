#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py into a per-forward table (steady state).
usage: prof_summary.py <dir with *_kernel_trace.csv> [marker-kernel-substring] [launches-per-forward]"""
import collections
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
lo, hi = gi[per_fwd * (nf - 2)], gi[per_fwd * (nf - 1)]          # one full steady-state forward period
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi]["Start_Timestamp"])
per, cnt = collections.defaultdict(float), collections.Counter()
for r in rows[lo:hi]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:80]
    per[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[n] += 1
busy = sum(per.values())
print(f"forwards seen {nf}; one forward period {(t1 - t0) / 1e6:.3f} ms, kernel busy {busy / 1e6:.3f} ms, launches {hi - lo}")
for n, v in sorted(per.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{v / 1e3:9.1f} us  x{cnt[n]:4d}  avg {v / 1e3 / cnt[n]:8.1f} us  {n}")
