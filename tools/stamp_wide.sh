#!/usr/bin/env bash
# diagnostic build of the wide-stage kernels (csrc/ms_wide.hip) with in-kernel cycle stamps: where does a workgroup's time go
# (prologue / main loop / fp32 epilogue / neuron epilogue + stores), and how do the workgroups of a launch spread in time?
# The diagnostic library lives beside, not over, the product one.   usage (GPU box): tools/stamp_wide.sh [B D H W C]
# SDF_STAMP_LIB=path: use a diagnostic library built beforehand (tools/stamp_wide.sh build NAME [flags] -> build/stamp/libNAME.so; the
# hipcc cross-compile runs off the GPU box and the .so travels with the snapshot) instead of compiling on the box.
set -e
cd "$(dirname "$0")/.."
build() {   # $1 = output .so, rest = extra flags
  local out=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_STAMP ${SDF_EXTRA_FLAGS:-} "$@" -c ${SDF_WIDE_SRC:-sdformerflow_amd/csrc/ms_wide.hip} -o ${out%.so}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out ${out%.so}.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v ms_wide)
}
if [ "$1" = build ]; then mkdir -p build/stamp; name=$2; shift 2; build build/stamp/lib$name.so "$@"; echo build/stamp/lib$name.so; exit 0; fi
if [ -z "$SDF_STAMP_LIB" ]; then build /tmp/libsdf_stamp_wide.so; SDF_STAMP_LIB=/tmp/libsdf_stamp_wide.so; fi
SDF_HIP_LIB=$SDF_STAMP_LIB python3 - "$@" <<'PY'
import ctypes, sys, os, torch
import numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from sdformerflow_amd import hip
import wide_one
conv = len(sys.argv) > 1 and sys.argv[1] == "conv"
if conv:                                     # the U-Net bottleneck's 3x3 convolution: stamp_wide.sh conv [B T H W Cin Cout]
    a = [int(v) for v in sys.argv[2:8]] if len(sys.argv) >= 8 else [1, 10, 9, 12, 768, 768]
    run = wide_one.conv(*a)
else:
    a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [1, 10, 18, 24, 384]
    run = wide_one.block(*a)
for _ in range(20):
    run()
torch.cuda.synchronize()
b = (ctypes.c_ulonglong * 48)()
c = (ctypes.c_ulonglong * (5 * 2048))()
hip.lib().sdf_debug_read_stamps_wide(b, c)
print("shape", a)
lp = b[40:48]
for kind, name in (((4, "conv K-split"),) if conv else ((0, "front"), (3, "proj+SN1"), (1, "fc1"), (2, "fc2"))):
    o = b[8 * kind:8 * kind + 8]
    clk = o[4] / max(o[5], 1) * 100e6 / 1e9
    g = int(o[6])
    arr = np.array(c[kind * 2048:kind * 2048 + 2 * min(g, 1024)], dtype=np.int64).reshape(-1, 2)
    arr = arr[arr[:, 1] > 0]
    base = arr[:, 0].min()
    print(f"{name:9s} grid {g:4d}: prologue {o[0]:6d}  main loop {o[1]:6d}  fp32 epilogue {o[2]:6d}  neuron + stores {o[3]:6d}  total {o[4]:6d} cycles = {o[5] / 100:.2f} us at {clk:.2f} GHz"
          f" | launch: first start -> last end {(arr[:, 1].max() - base) / 100:.2f} us, starts spread {(arr[:, 0].max() - base) / 100:.2f} us, mean life {(arr[:, 1] - arr[:, 0]).mean() / 100:.2f} us, max life {(arr[:, 1] - arr[:, 0]).max() / 100:.2f} us")
print(f"main loop of the last launch (fc2), thread 0 of the middle workgroup, {lp[4]} chunks: weight commit + request {lp[0]}  MFMA chunk {lp[1]}  "
      f"operand copy + request {lp[2]}  barrier {lp[3]} cycles over chunks 1..; fill: issue {lp[5] >> 32}  first weights landed + committed {lp[5] & 0xFFFFFFFF}  "
      f"second requests + barrier {lp[6]}  chunk 0 whole {lp[7]}")
PY

