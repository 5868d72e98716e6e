#!/usr/bin/env python3
"""Cross-check of bench.py's live `roofline.us_per_launch` against the rocprofv3 kernel trace of the same command:
`time_dominant_kernels` launches the roofline shape (3x3 spike conv 96 -> 96 at 10 x 144 x 192) back to back, so its launches
are the longest run of consecutive dispatches of that kernel in the trace (the template itself also serves other shapes, which
is why the --stats average over the template is not the number to compare).  usage: prof_roofline_check.py <prof dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
for key, label in (("spike_mm_pp_kernel<2, 0, true>", "spike conv (roofline)"), ("neuron_kernel<10>", "neuron (roofline_neuron)")):
    best, cur = [], []
    for r in rows:
        if key in r["Kernel_Name"]:
            cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        else:
            if len(cur) > len(best):
                best = cur
            cur = []
    if len(cur) > len(best):
        best = cur
    timed = best[3:] if len(best) > 3 else best            # the first launches of the run are the untimed warm-up
    print(f"{label}: longest back-to-back run = {len(best)} launches; average of the timed ones {sum(timed)/len(timed):.1f} us "
          f"(min {min(timed):.1f}, max {max(timed):.1f})")
