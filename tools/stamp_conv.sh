#!/usr/bin/env bash
# diagnostic build with in-kernel cycle stamps (workgroup 0): where does a stage go, per role?  Run on the GPU box.
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_STAMP -c sdformerflow_amd/csrc/spike_mm_ws.hip -o /tmp/ws_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsdf_stamp.so /tmp/ws_stamp.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v spike_mm_ws)
# the diagnostic library lives beside, not over, the product one
SDF_HIP_LIB=/tmp/libsdf_stamp.so python3 - "$@" <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
sys.argv = ["conv_one.py"] + sys.argv[1:]
exec(open("tools/conv_one.py").read())
from sdformerflow_amd import hip
buf = (ctypes.c_ulonglong * 16)()
hip.lib().sdf_debug_read_stamps(buf)
Q = buf[3]
print(f"producer wave 4 (wg 0), Q={Q} stages: store {buf[0]/Q:.0f}  load-issue {buf[1]/Q:.0f}  barrier-wait {buf[2]/Q:.0f} cycles/stage")
print(f"consumer wave 0 (wg 0), Q={buf[7]}: mfma {buf[4]/Q:.0f}  epilogue {buf[5]/Q:.0f}  barrier-wait {buf[6]/Q:.0f} cycles/stage")
PY
