#!/usr/bin/env bash
# diagnostic build of the one-launch MS MLP with in-kernel cycle stamps (one workgroup in the middle of the grid, wave 0):
# where does a work item go?  The diagnostic library lives beside, not over, the product one.
# usage (GPU box): tools/stamp_mlp.sh [B D H W C]
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_STAMP ${SDF_EXTRA_FLAGS:-} -c sdformerflow_amd/csrc/ms_mlp_fused.hip -o /tmp/mlp_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsdf_stamp.so /tmp/mlp_stamp.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v ms_mlp_fused)
SDF_HIP_LIB=/tmp/libsdf_stamp.so python3 - "$@" <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from sdformerflow_amd import hip
import mlp_bench
a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [1, 10, 72, 96, 96]
mlp_bench.run(*a, reps=20, only_fused=True)
b = (ctypes.c_ulonglong * 16)()
hip.lib().sdf_debug_read_stamps_mlp(b)
n = max(b[7], 1)
clk = b[8] / max(b[9], 1) * 100e6 / 1e9
print(f"item of one workgroup (wave 0): SN1 + first weights {b[0]}  | per chunk: fc1 {b[1]/n:.0f}  BN1+SN2 {b[2]/n:.0f}  barrier+W2 store {b[3]/n:.0f}  fc2 {b[4]/n:.0f}  "
      f"barrier+W1 store {b[5]/n:.0f} | BN2 + shortcut {b[6]} | {n} chunks, item {b[8]} cycles, clock {clk:.2f} GHz")
occ = ctypes.c_int(0)
hip.lib().sdf_debug_occupancy_mlp(ctypes.byref(occ))
print("hipOccupancyMaxActiveBlocksPerMultiprocessor (T = 10, C = 96 instantiation):", occ.value)
# census of the LAST launch: how many workgroups were resident at once, and per compute unit
import numpy as np
c = (ctypes.c_ulonglong * (3 * 8192))()
hip.lib().sdf_debug_read_census_mlp(c)
B_, D_, H_, W_, C_ = a
per = (2 * (20 // D_) * 4 if C_ == 96 else (20 // D_) * 4) * 2; items = min(8192, (B_ * H_ * W_ + per - 1) // per)
arr = np.array(c[:3 * items], dtype=np.uint64).reshape(items, 3)
t0, t1, hw = arr[:, 0].astype(np.int64), arr[:, 1].astype(np.int64), arr[:, 2]
base = t0.min()
ev = sorted([(int(x - base), 1) for x in t0] + [(int(x - base), -1) for x in t1])
cur = peak = 0
for _, d in ev:
    cur += d; peak = max(peak, cur)
cu = ((hw >> np.uint64(32)) << np.uint64(16)) | ((hw & np.uint64(0xFFFFFFFF)) >> np.uint64(8) & np.uint64(0xFFFF))     # xcc | (se, sh, cu) bits of HW_ID
print(f"census: {items} workgroups, span {(t1.max() - base) / 100:.1f} us, mean life {(t1 - t0).mean() / 100:.1f} us, peak resident {peak}, "
      f"distinct (xcc, cu-ish) ids {len(set(cu.tolist()))}, first starts spread {(np.sort(t0)[min(items, 256) - 1] - base) / 100:.1f} us")
PY
