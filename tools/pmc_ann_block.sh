#!/usr/bin/env bash
# PMC counters of the one-launch attention half block (csrc/ann_block.hip) (own --pmc passes, no tracing besides the kernel list).
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
rm -rf gpurun_out/pmc_ann_block
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  timeout 120 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_ann_block -- python3 tools/ann_block_one.py "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_ann_block/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("ann_attn_block" in r["Kernel_Name"] or "ann_mlp_block" in r["Kernel_Name"]):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
