#!/usr/bin/env python3
"""Cross-check of bench.py's live roofline timings against the rocprofv3 kernel trace of the same command:
`time_dominant_kernels` launches each roofline form back to back (40 launches rotating through 4 operand sets, then 40 on one
set), so its launches are the long runs of consecutive dispatches of one kernel in the trace (the --stats average over a
kernel name mixes the forward's launches under three-way overlap with these).  Prints every run of >= 30 consecutive
launches of the roofline kernels in trace order: the digit convolution's membrane+spikes form (HBM-rotating, L3-resident),
its spikes-only form (same two), the fp32-epilogue form (same two), then the neuron kernel (HBM, L3).
usage: prof_roofline_check.py <prof dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
keys = ("spike_conv_wres_i8_kernel<10", "spike_conv_wres_i8_kernel<0", "spike_mm_pp_kernel<2, 10, true>", "spike_mm_pp_kernel<2, 0, true>",
        "neuron_kernel<10>")
run_key, cur = None, []


# bench._timed: max(3, number of operand sets) untimed launches, then 40 timed ones; rotating (4 conv sets / 7 neuron sets) first,
# then one set.  A run of back-to-back launches of one kernel is cut into those segments.
SEGMENTS = {"spike_conv_wres_i8_kernel<10": [("membrane+spikes, HBM-rotating", 4), ("membrane+spikes, L3-resident", 3),
                                             ("spikes only, HBM-rotating", 4), ("spikes only, L3-resident", 3)],
            "spike_conv_wres_i8_kernel<0": [("fp32 epilogue, HBM-rotating", 4), ("fp32 epilogue, L3-resident", 3)],
            "neuron_kernel<10>": [("HBM-rotating", 7), ("L3-resident", 3)]}


def flush():
    if not run_key or len(cur) < 30:
        return
    segs = SEGMENTS.get(run_key)
    if segs and len(cur) == sum(w + 40 for _, w in segs):
        i = 0
        for label, w in segs:
            timed = cur[i + w:i + w + 40]
            i += w + 40
            print(f"{run_key:32s} {label:30s}: 40 timed launches, average {sum(timed) / 40:7.1f} us (min {min(timed):.1f}, max {max(timed):.1f})")
    else:
        timed = cur[-40:]
        print(f"{run_key:32s} run of {len(cur)} back-to-back launches: average of the last 40 {sum(timed) / len(timed):7.1f} us")


for r in rows:
    k = next((k for k in keys if k in r["Kernel_Name"]), None)
    if k != run_key:
        flush()
        run_key, cur = k, []
    if k:
        cur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
flush()
