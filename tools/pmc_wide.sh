#!/usr/bin/env bash
# PMC counters of the wide-stage kernels (csrc/ms_wide.hip) from the one-block driver tools/wide_one.py, started directly behind
# `rocprofv3 ... --` (no shell / env hop); counters in separate --pmc passes; durations from a --kernel-trace pass.  Per kernel:
# instruction mix, MFMA-pipe busy share, LDS bank-conflict share, L2 hit rate, bytes through the L2's memory side (FETCH_SIZE x 2
# on gfx950, WRITE_SIZE).   usage (GPU box): tools/pmc_wide.sh [tag] [B D H W C]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
TAG=${1:-r4}; shift
ARGS=${@:-1 10 18 24 384}
OUT=gpurun_out/pmc_wide_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
for i in 0 1 2 3 4 5 6; do
  case $i in
    0) set_="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE";;
    1) set_="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM";;
    2) set_="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM";;
    3) set_="FETCH_SIZE";;
    4) set_="WRITE_SIZE";;
    5) set_="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum";;
    6) set_="";;
  esac
  if [ -n "$set_" ]; then
    timeout 180 rocprofv3 --pmc $set_ --output-format csv -d ${OUT:?}/p$i -- python3 tools/wide_one.py $ARGS > /dev/null 2>&1
  else
    timeout 180 rocprofv3 --kernel-trace --output-format csv -d ${OUT:?}/trace -- python3 tools/wide_one.py $ARGS > /dev/null 2>&1
  fi
done
python3 - "$OUT" "$ARGS" <<'PY' | tee gpurun_out/pmc_wide_$TAG.txt
import csv, glob, sys, collections
d, args = sys.argv[1:3]
print("tools/pmc_wide.sh: one MS block at (B D H W C) =", args)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for kn in sorted(acc):
    if "wide_" not in kn and "res_" not in kn and "neuron_kernel" not in kn and "smallm_" not in kn:
        continue
    m = {k: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for k, v in acc[kn].items()}      # later launches: warm
    dd = dur.get(kn, [])
    us = sorted(dd)[len(dd) // 2] if dd else float("nan")
    g = m.get
    print(f"== {kn[:120]}\n   duration (median of {len(dd)} launches, --kernel-trace pass) {us:.1f} us")
    if g("SQ_WAVES"):
        w = g("SQ_WAVES")
        print(f"   waves {w:.0f}; per wave: VALU {g('SQ_INSTS_VALU', 0) / w:.0f}  MFMA {g('SQ_INSTS_MFMA', 0) / w:.0f}  LDS {g('SQ_INSTS_LDS', 0) / w:.0f}  "
              f"SALU {g('SQ_INSTS_SALU', 0) / w:.0f}  VMEM rd {g('SQ_INSTS_VMEM_RD', 0) / w:.1f} wr {g('SQ_INSTS_VMEM_WR', 0) / w:.1f}  SMEM {g('SQ_INSTS_SMEM', 0) / w:.0f}"
              f"   -> VALU per MFMA {g('SQ_INSTS_VALU', 0) / max(g('SQ_INSTS_MFMA', 1), 1):.2f} (16x16x32 MFMAs: two per 32x32x16-equivalent)")
        print(f"   wave cycles: busy {g('SQ_BUSY_CYCLES', 0):.3e}  wave {g('SQ_WAVE_CYCLES', 0):.3e}  wait_any {g('SQ_WAIT_ANY', 0):.3e}  wait_inst {g('SQ_WAIT_INST_ANY', 0):.3e}  active_inst {g('SQ_ACTIVE_INST_ANY', 0):.3e}")
    if g("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE / 8 counts from the counter window's opening to its closing: on launches of tens of microseconds that is more
        # than the kernel (round 4's tables showed 3 - 4.7 "GHz" on a 2.4 GHz part and understated every busy share by up to 2 x,
        # VERDICT r4 weak #9).  The kernel's cycles are its own duration x the clock it ran at, the clock read from the quotient but
        # never above the part's 2.4 GHz.
        quot = g("GRBM_GUI_ACTIVE") / 8 / (us * 1e-6) / 1e9
        clk = min(quot, 2.4)
        cyc = us * 1e-6 * clk * 1e9
        print(f"   kernel cycles {cyc:.3e} = {us:.1f} us x {clk:.2f} GHz (GRBM_GUI_ACTIVE / 8 / duration reads {quot:.2f} GHz); SQ_VALU_MFMA_BUSY_CYCLES "
              f"{g('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3e} -> matrix pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc):.3f} of SIMD-cycles")
    if g("SQ_LDS_IDX_ACTIVE"):
        print(f"   LDS bank conflicts {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.3f} of LDS cycles")
    if g("TCC_REQ_sum"):
        print(f"   L2: hit {g('TCC_HIT_sum', 0):.3e} miss {g('TCC_MISS_sum', 0):.3e} -> hit rate {g('TCC_HIT_sum', 0) / max(g('TCC_HIT_sum', 0) + g('TCC_MISS_sum', 0), 1):.3f}; requests {g('TCC_REQ_sum'):.3e} (reads {g('TCC_READ_sum', 0):.3e})")
    if g("FETCH_SIZE") is not None:
        print(f"   memory side of L2 per launch: FETCH_SIZE {g('FETCH_SIZE'):.1f} KB x 2 (gfx950) = {2 * g('FETCH_SIZE') / 1e3:.2f} MB read, WRITE_SIZE {g('WRITE_SIZE', 0) / 1e3:.2f} MB written")
PY
rm -rf "$OUT"
