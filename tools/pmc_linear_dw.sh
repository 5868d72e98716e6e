#!/usr/bin/env bash
# the weight-gradient kernel (csrc/linear_dw.hip): duration from a kernel trace, then matrix-pipe / LDS / HBM counters in their own
# --pmc passes (FETCH_SIZE / WRITE_SIZE alone, as MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 on gfx950).
# usage (GPU box): tools/pmc_linear_dw.sh 276480 96 96   |   tools/pmc_linear_dw.sh conv 40 96 144 192
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
rm -rf gpurun_out/pmc_ldw
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pmc_ldw/trace -o t -- python3 tools/linear_dw_one.py "$@" > /dev/null 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_ldw/pmc -- python3 tools/linear_dw_one.py "$@" > /dev/null 2>&1
done
python3 - "$@" <<'PY'
import csv, glob, collections, sys
print("tools/pmc_linear_dw.sh", " ".join(sys.argv[1:]))
f = glob.glob("gpurun_out/pmc_ldw/trace/**/*_kernel_stats.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("linear_dw", "ringed_rows")):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e3
        print(f"  {r['Name'][:90]:90s} calls {r['Calls']:>4s}  avg {float(r['AverageNs'])/1e3:8.1f} us")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ldw/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "linear_dw_kernel" in r["Kernel_Name"]:
            acc["linear_dw_kernel"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kern, cs in acc.items():
    m = {k: sum(v) / len(v) for k, v in cs.items()}
    for k in sorted(m): print(f"  {k:28s} {m[k]:16.0f}")
    us = next((v for k, v in dur.items() if "linear_dw_kernel" in k), 0.0)
    cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8                       # the counter sums the 8 XCDs
    clk = min(cyc / us / 1e3, 2.4) if us else 0.0
    print(f"  -> {us:.1f} us per launch at {clk:.2f} GHz; matrix pipe busy {100 * m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(us * clk * 1e3 * 1024, 1):.1f} % of 1 024 SIMDs; "
          f"VALU per MFMA {m.get('SQ_INSTS_VALU', 0) / max(m.get('SQ_INSTS_MFMA', 1), 1):.1f}; LDS conflict cycles {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1):.1f} % of LDS active; "
          f"HBM read {2 * m.get('FETCH_SIZE', 0) / 1024:.1f} MB (FETCH_SIZE x 2) + write {m.get('WRITE_SIZE', 0) / 1024:.1f} MB per launch")
PY
rm -rf gpurun_out/pmc_ldw
