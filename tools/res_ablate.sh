#!/usr/bin/env bash
# What bounds the weight-resident row-loop kernel on a plain product?  Diagnostic builds of csrc/ms_res.hip with one thing removed each
# (the spike operand's loads read nothing / no matrix instruction / no fp32 stores) beside the product build, timed by a rocprofv3 kernel
# trace of tools/res_gemm_one.py.   usage: tools/res_ablate.sh build (off the GPU box) ; tools/res_ablate.sh [M N K] (GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/ablate
  for v in base NOA NOMFMA NOST; do
    fl=""; [ $v != base ] && fl="-DRES_X_$v"
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -Wno-unused-function $fl -c sdformerflow_amd/csrc/ms_res.hip -o build/ablate/ms_res_$v.o &
  done
  wait
  for v in base NOA NOMFMA NOST; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/libres_$v.so build/ablate/ms_res_$v.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v "ms_res.o")
  done
  ls build/ablate/*.so; exit 0
fi
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in base NOA NOMFMA NOST; do
  rm -rf "$R/gpurun_out/abl_$v"
  SDF_HIP_LIB=$R/build/ablate/libres_$v.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/abl_$v" -o t -- python3 "$R/tools/res_gemm_one.py" "$@" > /dev/null 2>&1
  python3 - "$R/gpurun_out/abl_$v" $v <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f)) if "res_pm_kernel" in r["Kernel_Name"])
print(f"{sys.argv[2]:8s} res_pm_kernel: median {d[len(d) // 2]:.1f} us of {len(d)} launches (min {d[0]:.1f})")
PY
  rm -rf "$R/gpurun_out/abl_$v"
done
