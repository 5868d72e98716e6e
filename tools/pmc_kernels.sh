#!/usr/bin/env bash
# PMC counters of the kernels that carry the headline, each from its own one-shot driver started directly
# behind `rocprofv3 ... --` (no shell / env hop), counters in separate --pmc passes (slot limits; FETCH_SIZE and WRITE_SIZE apart),
# durations from a --kernel-trace pass.  Prints, per kernel: instruction mix (VALU / MFMA / LDS per wave), MFMA-pipe busy share,
# LDS bank-conflict share, effective clock (GRBM_GUI_ACTIVE / 8 / duration) and HBM bytes (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE).
# usage (GPU box): tools/pmc_kernels.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
TAG=${1:-r3}
OUT=gpurun_out/pmc_$TAG
rm -rf ${OUT:?}; mkdir -p ${OUT:?}
run_one() {   # name, kernel substring, program args...
  local name=$1 sub=$2; shift 2
  for i in 0 1 2 3 4 5; do
    case $i in
      0) set_="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE";;
      1) set_="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM";;
      2) set_="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM";;
      3) set_="FETCH_SIZE";;
      4) set_="WRITE_SIZE";;
      5) set_="";;
    esac
    if [ -n "$set_" ]; then
      timeout 180 rocprofv3 --pmc $set_ --output-format csv -d ${OUT:?}/$name/p$i -- python3 "$@" > /dev/null 2>&1
    else
      timeout 180 rocprofv3 --kernel-trace --output-format csv -d ${OUT:?}/$name/trace -- python3 "$@" > /dev/null 2>&1
    fi
  done
  python3 - "${OUT:?}/$name" "$sub" "$name" <<'PY'
import csv, glob, sys, collections
d, sub, name = sys.argv[1:4]
acc = collections.defaultdict(list)
kn = None
for f in glob.glob(d + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            kn = r["Kernel_Name"]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = []
for f in glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
m = {k: sum(v[len(v) // 2:]) / max(len(v[len(v) // 2:]), 1) for k, v in acc.items()}      # later launches: warm
us = sorted(dur)[len(dur) // 2] if dur else float("nan")
print(f"== {name}: {(kn or sub)[:110]}")
print(f"   duration (median of {len(dur)} launches, --kernel-trace pass) {us:.1f} us")
g = m.get
if g("SQ_WAVES"):
    w = g("SQ_WAVES")
    print(f"   waves {w:.0f}; per wave: VALU {g('SQ_INSTS_VALU', 0) / w:.0f}  MFMA {g('SQ_INSTS_MFMA', 0) / w:.0f}  LDS {g('SQ_INSTS_LDS', 0) / w:.0f}  "
          f"SALU {g('SQ_INSTS_SALU', 0) / w:.0f}  VMEM rd {g('SQ_INSTS_VMEM_RD', 0) / w:.1f} wr {g('SQ_INSTS_VMEM_WR', 0) / w:.1f}  SMEM {g('SQ_INSTS_SMEM', 0) / w:.0f}"
          f"   -> VALU per MFMA {g('SQ_INSTS_VALU', 0) / max(g('SQ_INSTS_MFMA', 1), 1):.2f}")
if g("GRBM_GUI_ACTIVE"):
    quot = g("GRBM_GUI_ACTIVE") / 8 / (us * 1e-6) / 1e9
    clk = min(quot, 2.4)                 # (the quotient reads high on launches under 0.3 ms: the counter window is longer than the kernel)
    print(f"   GRBM_GUI_ACTIVE {g('GRBM_GUI_ACTIVE'):.0f} (sum over 8 XCDs) / 8 / duration = {quot:.2f} GHz -> clock taken {clk:.2f} GHz (never above the part's 2.4)")
    if g("SQ_VALU_MFMA_BUSY_CYCLES"):
        # per-SIMD busy cycles summed over the chip's 1024 SIMDs; kernel cycles = the kernel's own duration x the clock
        cyc = us * 1e-6 * clk * 1e9
        print(f"   SQ_VALU_MFMA_BUSY_CYCLES {g('SQ_VALU_MFMA_BUSY_CYCLES'):.3e}; SQ_BUSY_CYCLES {g('SQ_BUSY_CYCLES', 0):.3e}; SQ_WAVE_CYCLES {g('SQ_WAVE_CYCLES', 0):.3e}; "
              f"kernel cycles {cyc:.3e}: MFMA busy / (256 CUs x kernel cycles) = {g('SQ_VALU_MFMA_BUSY_CYCLES') / (256 * cyc):.2f} (x4 if the counter is per SIMD)")
if g("SQ_LDS_IDX_ACTIVE"):
    print(f"   LDS: SQ_LDS_BANK_CONFLICT {g('SQ_LDS_BANK_CONFLICT', 0):.3e} of SQ_LDS_IDX_ACTIVE {g('SQ_LDS_IDX_ACTIVE'):.3e} = {g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.3f}")
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
    print(f"   HBM per launch: FETCH_SIZE {g('FETCH_SIZE'):.1f} KB x 2 (gfx950) = {2 * g('FETCH_SIZE') / 1e3:.1f} MB read, WRITE_SIZE {g('WRITE_SIZE') / 1e3:.1f} MB written "
          f"(KB = 1000 B as rocprofv3 reports; KiB if the tool means 1024: x1.024)")
for k in sorted(m):
    print(f"      {k:30s} {m[k]:18.0f}")
PY
}
run_one conv_fusedm   "spike_conv_wres_i8_kernel"  tools/conv_one.py 10 144 192 96 96 1 fusedm i8x3
run_one conv_fused    "spike_conv_wres_i8_kernel"  tools/conv_one.py 10 144 192 96 96 1 fused i8x3
run_one mlp_stage0    "ms_mlp_fused_kernel"        tools/mlp_one.py 1 10 72 96 96
run_one gemm_proj_s2  "spike_gemm_kernel"          tools/gemm_one.py 4860 384 384 0 3
run_one gemm_fc1_s2   "spike_mm_pp_kernel"         tools/gemm_one.py 4320 1536 384 10 1
run_one attn_ann_s0   "win_attn_tiled_f16_kernel"  tools/win_attn_one.py ann 704 3 162 mask
rm -rf ${OUT:?}/*/p* ${OUT:?}/*/trace
