#!/usr/bin/env bash
# Counters of BASELINE configs[3]'s training step (local batch 4), per kernel family: vector / matrix instructions, matrix-pipe busy
# cycles, HBM bytes (separate --pmc passes; FETCH_SIZE x 2 on gfx950, KiB) summed over the launches of ONE steady-state step of `python3
# bench.py --train --steps 3 --warmup 1`: from the gradient clip of the third step (its one `lpnorm_cleanup` launch) to the clip of the
# fourth - optimiser update, forward, loss, backward; MIOpen's one-time kernel search of the first step stays outside.
# usage (GPU box): tools/pmc_train_step.sh [tag] -> gpurun_out/<tag>_pmc_train_step.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
TAG=${1:-r6}
OUT=gpurun_out/pmct_$TAG
rm -rf ${OUT:?}; mkdir -p ${OUT:?}
i=0
for set_ in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 600 rocprofv3 --pmc $set_ --output-format csv -d ${OUT:?}/p$i -- python3 bench.py --train --local-batch 4 --steps 3 --warmup 1 > /dev/null 2>&1 < /dev/null
  i=$((i+1))
done
python3 - ${OUT:?} <<'PY' > gpurun_out/${TAG}_pmc_train_step.txt
import csv, glob, sys, collections, re
out = sys.argv[1]
def fam(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("sdfmm::", "")
    n = re.sub(r"\(.*", "", n)
    if "miopenSp3AsmConv" in n: return "MIOpen Winograd conv"
    if n.startswith("igemm_"): return "MIOpen implicit-GEMM conv (" + n.split("_")[1] + ")"
    if n.startswith("Cijk_"): return "rocBLAS GEMM"
    if "batched_transpose" in n or "transpose" in n.lower(): return "MIOpen / ATen transposes"
    if n.startswith("at::native::") or n.startswith("at::"): return "ATen " + re.sub(r"<.*", "", n.split("::")[-1])[:40]
    return re.sub(r"<.*", "", n)[:60]
per = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for d in sorted(glob.glob(out + "/p*")):
    rows = [r for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
    clips = sorted({int(r["Dispatch_Id"]) for r in rows if "lpnorm_cleanup" in r["Kernel_Name"]})
    lo, hi = clips[-2], clips[-1]
    for r in rows:
        if lo < int(r["Dispatch_Id"]) <= hi:
            k = fam(r["Kernel_Name"])
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if d.endswith("p0") and r["Counter_Name"] == "SQ_WAVES":
                cnt[k] += 1
S = 1.0
tot = collections.defaultdict(float)
for k, v in per.items():
    for c, x in v.items():
        tot[c] += x
print(f"configs[3] training step, local batch 4, one steady-state step ({sum(cnt.values())} launches): VALU {tot['SQ_INSTS_VALU'] / S / 1e6:.0f} M wave-instructions "
      f"({tot['SQ_INSTS_VALU'] / S * 4 / 1024 / 2.4e6:.1f} ms of the chip's vector issue at 2.4 GHz), MFMA {tot['SQ_INSTS_MFMA'] / S / 1e6:.1f} M, matrix pipe busy "
      f"{tot['SQ_VALU_MFMA_BUSY_CYCLES'] / S / 1024 / 2.4e6:.1f} ms per SIMD, HBM {2 * tot['FETCH_SIZE'] * 1024 / S / 1e9:.1f} GB read + {tot['WRITE_SIZE'] * 1024 / S / 1e9:.1f} GB written "
      f"({(2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) * 1024 / S / 8e9:.1f} ms at 8 TB/s)")
print(f"{'kernel family':62s} {'n/step':>7s} {'VALU M':>8s} {'MFMA M':>8s} {'pipe ms':>8s} {'read GB':>8s} {'write GB':>8s} {'HBM ms':>7s}")
key = lambda k: -(2 * per[k]['FETCH_SIZE'] + per[k]['WRITE_SIZE'])
for k in sorted(per, key=key)[:40]:
    v = per[k]
    print(f"{k:62s} {cnt[k] / S:7.1f} {v['SQ_INSTS_VALU'] / S / 1e6:8.1f} {v['SQ_INSTS_MFMA'] / S / 1e6:8.2f} {v['SQ_VALU_MFMA_BUSY_CYCLES'] / S / 1024 / 2.4e6:8.2f} "
          f"{2 * v['FETCH_SIZE'] * 1024 / S / 1e9:8.2f} {v['WRITE_SIZE'] * 1024 / S / 1e9:8.2f} {(2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024 / S / 8e9:7.2f}")
PY
rm -rf ${OUT:?}
cat gpurun_out/${TAG}_pmc_train_step.txt | head -50
