#!/usr/bin/env bash
# Where one forward of BASELINE config 2 spends the chip: per kernel name, summed over the launches of the last of five eager
# forwards - vector / matrix instruction counts, matrix-pipe busy cycles, wave-cycles, HBM bytes (separate --pmc passes; FETCH_SIZE
# x 2 on gfx950, KiB) - and the kernel durations of a --kernel-trace pass.  usage (GPU box): tools/pmc_forward2.sh [tag] [lif|psn] [R]
# R > 1: one launch sequence over R samples (forward_replicas); the table is per launch sequence, gpurun_out/<tag>_kernel_counters.json
# (what bench.py's `roofline.by_kernel` reads from profiles/) is per SAMPLE.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
TAG=${1:-r3}
KIND=${2:-lif}
REP=${3:-1}
OUT=gpurun_out/pmcf_$TAG
rm -rf ${OUT:?}; mkdir -p ${OUT:?}
i=0
for set_ in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
            "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $set_ --output-format csv -d ${OUT:?}/p$i -- python3 tools/forward_one.py 5 $KIND $REP > /dev/null 2>&1 < /dev/null
  i=$((i+1))
done
timeout 300 rocprofv3 --kernel-trace --output-format csv -d ${OUT:?}/trace -- python3 tools/forward_one.py 5 $KIND $REP > /dev/null 2>&1 < /dev/null
python3 - ${OUT:?} $REP $KIND <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("sdfmm::", "")
    return re.sub(r"\(.*", "", n)[:64]
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in sorted(glob.glob(out + "/p*")):
    rows = [r for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
    names = sorted({r["Counter_Name"] for r in rows})
    byd = collections.defaultdict(dict)
    for r in rows:
        byd[int(r["Dispatch_Id"])][r["Counter_Name"]] = (float(r["Counter_Value"]), r["Kernel_Name"])
    ids = sorted(byd)
    heads = [i for i in ids if "head_conv_" in next(iter(byd[i].values()))[1]]
    lo, hi = heads[-2], heads[-1]                       # the launches from the 4th forward's head convolution to the 5th's
    for i in ids:
        if lo <= i < hi:
            for c, (v, k) in byd[i].items():
                per[short(k)][c] += v
            if d.endswith("p0"):
                cnt[short(next(iter(byd[i].values()))[1])] += 1
dur = collections.defaultdict(float)
rows = sorted((r for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))), key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "head_conv_" in r["Kernel_Name"]]
for r in rows[heads[-2]:heads[-1]]:
    dur[short(r["Kernel_Name"])] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = collections.defaultdict(float)
for k, v in per.items():
    for c, x in v.items():
        tot[c] += x
R = int(sys.argv[2])
import json
json.dump({"source": "tools/pmc_forward2.sh (rocprofv3 --pmc, separate passes; FETCH_SIZE x 2, KiB) on one eager launch sequence", "neuron": sys.argv[3],
           "samples_per_launch_sequence": R, "per_sample": {k: {"launches": cnt[k], "us": dur[k] / R, "valu_wave_insts": v["SQ_INSTS_VALU"] / R,
           "mfma_insts": v["SQ_INSTS_MFMA"] / R, "mfma_busy_cycles": v["SQ_VALU_MFMA_BUSY_CYCLES"] / R, "hbm_read_bytes": 2 * v["FETCH_SIZE"] * 1024 / R,
           "hbm_write_bytes": v["WRITE_SIZE"] * 1024 / R, "lds_conflict_frac": v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)} for k, v in per.items()}},
          open(out.replace("pmcf_", "") + "_kernel_counters.json", "w"), indent=1)
print(f"one eager launch sequence of config 2 over {R} sample(s): {sum(cnt.values())} launches, kernel time {sum(dur.values()):.0f} us; VALU {tot['SQ_INSTS_VALU'] / 1e6:.1f} M wave-instructions, "
      f"MFMA {tot['SQ_INSTS_MFMA'] / 1e6:.2f} M, matrix-pipe busy {tot['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / 1e3:.0f} k cycles per SIMD, "
      f"HBM {2 * tot['FETCH_SIZE'] * 1024 / 1e9:.2f} GB read + {tot['WRITE_SIZE'] * 1024 / 1e9:.2f} GB written")
print(f"{'kernel':64s} {'n':>3s} {'us':>7s} {'VALU M':>8s} {'MFMA M':>7s} {'VALU/MFMA':>9s} {'pipe busy %':>11s} {'LDS confl %':>11s} {'read MB':>8s} {'write MB':>8s} {'LDS M':>7s} {'LDS busy %':>10s}")
for k in sorted(per, key=lambda k: -dur[k]):
    v = per[k]
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / max(dur[k] * 2200, 1) * 100          # per SIMD, against the launch time at ~2.2 GHz
    print(f"{k:64s} {cnt[k]:3d} {dur[k]:7.1f} {v['SQ_INSTS_VALU'] / 1e6:8.2f} {v['SQ_INSTS_MFMA'] / 1e6:7.3f} {v['SQ_INSTS_VALU'] / max(v['SQ_INSTS_MFMA'], 1):9.1f} "
          f"{busy:11.1f} {100 * v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1):11.1f} {2 * v['FETCH_SIZE'] * 1024 / 1e6:8.1f} {v['WRITE_SIZE'] * 1024 / 1e6:8.1f} "
          f"{v['SQ_INSTS_LDS'] / 1e6:7.2f} {100 * v['SQ_LDS_IDX_ACTIVE'] / 1024 / max(dur[k] * 2200, 1):10.1f}")
PY
rm -rf ${OUT:?}/p* ${OUT:?}/trace
