#!/usr/bin/env bash
# diagnostic build of the weight-resident narrow-stage kernels (csrc/ms_res.hip) with in-kernel cycle stamps: where does a launch's time
# go (prologue requests / weights + barrier / first unit's main loop / its epilogue / the remaining units), and how do the workgroups
# spread in time?  The diagnostic library lives beside, not over, the product one.
# usage: tools/stamp_res.sh build           (off the GPU box: build/stamp/libres.so travels with the snapshot)
#        tools/stamp_res.sh [B D H W C]     (GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/stamp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -Wno-unused-function -DSDF_STAMP ${SDF_EXTRA_FLAGS:-} -c sdformerflow_amd/csrc/ms_res.hip -o build/stamp/ms_res.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamp/libres.so build/stamp/ms_res.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v ms_res)
  echo build/stamp/libres.so; exit 0
fi
SDF_HIP_LIB=${SDF_STAMP_LIB:-build/stamp/libres.so} python3 - "$@" <<'PY'
import ctypes, sys, os, torch
import numpy as np
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from sdformerflow_amd import hip
import wide_one
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [1, 10, 36, 48, 192]
run = wide_one.block(*a)
for _ in range(20):
    run()
torch.cuda.synchronize()
b = (ctypes.c_ulonglong * 32)()
c = (ctypes.c_ulonglong * (4 * 2048))()
hip.lib().sdf_debug_read_stamps_res(b, c)
print("tools/stamp_res.sh: one MS block at (B D H W C) =", a, "- wave 0 of the middle workgroup, cycles")
for kind, name in ((3, "proj+SN1"), (1, "fc1"), (2, "fc2")):
    o = b[8 * kind:8 * kind + 8]
    g = int(o[6])
    if g == 0:
        continue
    arr = np.array(c[kind * 2048:kind * 2048 + 2 * min(g, 1024)], dtype=np.int64).reshape(-1, 2)
    arr = arr[arr[:, 1] > 0]
    base = arr[:, 0].min()
    tot = sum(o[:5])
    print(f"{name:9s} grid {g:4d}, {o[7]} units per wave: entry -> first unit addressed {o[0]:6d} | -> weights in LDS + barrier {o[1]:6d} | first unit: main loop {o[2]:6d}  "
          f"epilogue {o[3]:6d} | remaining units + drain {o[4]:6d} | total {tot:6d} cycles = {o[5] / 100:.2f} us"
          f" || launch: first start -> last end {(arr[:, 1].max() - base) / 100:.2f} us, starts spread {(arr[:, 0].max() - base) / 100:.2f} us, "
          f"mean life {(arr[:, 1] - arr[:, 0]).mean() / 100:.2f} us, max life {(arr[:, 1] - arr[:, 0]).max() / 100:.2f} us")
PY
