#!/usr/bin/env bash
# HBM bytes (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, gfx950 x2 read correction) and duration of the digit convolution's two
# forms under both workgroup -> (tile range, column block) mappings: SDF_CONV_WRES_CB_INNER = 0 (column-block-major ranges, round 2)
# and 1 (the column blocks of a tile range side by side on one XCD).  usage (GPU box): tools/pmc_conv_ab.sh
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
OUT=gpurun_out/pmc_conv_ab
rm -rf ${OUT:?}; mkdir -p ${OUT:?}
for form in fused fusedm; do
  for cbi in 0 1; do
    export SDF_CONV_WRES_CB_INNER=$cbi
    d=${OUT:?}/${form}_$cbi
    timeout 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/p0 -- python3 tools/conv_one.py 10 144 192 96 96 1 $form i8x3 > /dev/null 2>&1
    timeout 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/p1 -- python3 tools/conv_one.py 10 144 192 96 96 1 $form i8x3 > /dev/null 2>&1
    timeout 180 rocprofv3 --kernel-trace --output-format csv -d $d/trace -- python3 tools/conv_one.py 10 144 192 96 96 1 $form i8x3 > /dev/null 2>&1
    python3 - $d $form $cbi <<'PY'
import csv, glob, sys, collections
d, form, cbi = sys.argv[1:4]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spike_conv_wres_i8_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spike_conv_wres_i8_kernel" in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for k, v in acc.items()}
us = sorted(dur)[len(dur) // 2] if dur else float("nan")
print(f"{form:7s} cb_inner={cbi}: {us:6.1f} us (median of {len(dur)}); FETCH_SIZE {m.get('FETCH_SIZE', 0):9.1f} KB x 2 = {2 * m.get('FETCH_SIZE', 0) * 1.024 / 1e3:6.1f} MB read, "
      f"WRITE_SIZE {m.get('WRITE_SIZE', 0) * 1.024 / 1e3:6.1f} MB written (KB = 1024 B)")
PY
  done
done | tee ${OUT:?}/summary.txt
rm -rf ${OUT:?}/*_0 ${OUT:?}/*_1
