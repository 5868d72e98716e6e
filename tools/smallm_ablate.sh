#!/usr/bin/env bash
# What bounds the small-M convolution kernel (csrc/ms_smallm.hip)?  Diagnostic builds with one part of the main loop removed, timed with
# rocprofv3 on the U-Net bottleneck shape.  Build here (hipcc cross-compiles; the .so files travel with the snapshot):
#   tools/smallm_ablate.sh build        -> build/smallm/lib{base,nomfma,noa,nob,nostrip}.so
# then on the GPU box:  tools/smallm_ablate.sh run [B T H W Cin Cout]
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/smallm
  for v in stamp:-DSDF_STAMP base: nomfma:-DSMX_NOMFMA noa:-DSMX_NOA nob:-DSMX_NOB nostrip:-DSMX_NOSTRIP noab:"-DSMX_NOA -DSMX_NOB"; do
    n=${v%%:*}; f=${v#*:}
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed $f -c sdformerflow_amd/csrc/ms_smallm.hip -o build/smallm/$n.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/smallm/lib$n.so build/smallm/$n.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v ms_smallm) ) &
  done
  wait
  ls build/smallm/*.so
  exit 0
fi
if [ "$1" = stamp ]; then       # in-kernel cycle stamps of the middle workgroup + every workgroup's life
  shift
  SDF_HIP_LIB=$PWD/build/smallm/libstamp.so python3 - "$@" <<'PY'
import ctypes, sys, os, torch
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from sdformerflow_amd import hip
import wide_one
a = [int(v) for v in sys.argv[1:7]] if len(sys.argv) >= 7 else [1, 10, 9, 12, 768, 768]
run = wide_one.conv(*a)
for _ in range(20):
    run()
torch.cuda.synchronize()
b = (ctypes.c_ulonglong * 32)(); c = (ctypes.c_ulonglong * 2048)()
hip.lib().sdf_debug_read_stamps_smallm(b, c)
print("shape", a, "grid", b[7])
for w in range(4):
    o = b[8 * w:8 * w + 8]
    print(f"wave {w}: prologue {o[0]:6d}  fill {o[1]:6d}... main loop(+fill) {o[1]:6d}  reduce {o[2]:6d}  epilogue {o[3]:6d}  tail {o[4]:6d}  total {o[5]:6d} cycles = {o[6] / 100:.2f} us -> {o[5] / max(o[6], 1) / 10:.2f} GHz")
g = int(b[7])
arr = np.array(c[:2 * min(g, 1024)], dtype=np.int64).reshape(-1, 2)
arr = arr[arr[:, 1] > 0]
base = arr[:, 0].min()
print(f"launch: first start -> last end {(arr[:, 1].max() - base) / 100:.2f} us, starts spread {(arr[:, 0].max() - base) / 100:.2f} us, "
      f"mean life {(arr[:, 1] - arr[:, 0]).mean() / 100:.2f} us, max life {(arr[:, 1] - arr[:, 0]).max() / 100:.2f} us, min life {(arr[:, 1] - arr[:, 0]).min() / 100:.2f} us")
PY
  exit 0
fi
shift || true
for n in base nomfma noa nob noab nostrip; do
  echo "== $n"
  SDF_HIP_LIB=$PWD/build/smallm/lib$n.so bash tools/prof_any.sh abl_$n tools/wide_one.py conv "$@" 2>&1 | grep smallm_kernel | sed 's/void sdfmm::(anonymous namespace):://; s/(sdfmm.*Params)//'
done
