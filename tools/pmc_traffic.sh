#!/usr/bin/env bash
# HBM traffic of the dominant spike-conv kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (TCC slots),
# as MI355X_MICROARCH.md prescribes.  FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950.
# usage: tools/pmc_traffic.sh [resid|fusedm|fused] [2|i8x3]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
rm -rf gpurun_out/pmc_traffic
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_traffic -- python3 tools/conv_one.py 10 144 192 96 96 1 ${1:-fusedm} ${2:-i8x3} > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_traffic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "spike_mm_pp" in r["Kernel_Name"] or "spike_conv_wres" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k}: mean {sum(v)/len(v):.1f} KB per launch over {len(v)} launches")
f = sum(acc["FETCH_SIZE"]) / max(len(acc["FETCH_SIZE"]), 1); w = sum(acc["WRITE_SIZE"]) / max(len(acc["WRITE_SIZE"]), 1)
print(f"HBM traffic per launch: read {2*f/1024:.1f} MB (FETCH_SIZE x2 correction) + write {w/1024:.1f} MB; "
      f"algorithmic: spikes in 26.5 MB + residual in 106.2 MB + fp32 out 106.2 MB (+ 26.5 MB spikes out when fused) + weights 0.3 MB")
PY
