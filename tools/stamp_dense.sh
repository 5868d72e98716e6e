#!/usr/bin/env bash
# diagnostic build of the dense 3x3 convolution with in-kernel cycle stamps (workgroup 0, waves 0 and 4): where does a
# step go?  The diagnostic library lives beside, not over, the product one.  usage (GPU box): tools/stamp_dense.sh [imgs H W]
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_STAMP -c sdformerflow_amd/csrc/dense_conv_wres.hip -o /tmp/dense_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsdf_stamp.so /tmp/dense_stamp.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v dense_conv_wres)
SDF_HIP_LIB=/tmp/libsdf_stamp.so python3 - "$@" <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
from sdformerflow_amd import hip
imgs, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 288, 384)
g = torch.Generator().manual_seed(0)
xp = hip.pack_planes(torch.randn(imgs, 96, H, W, generator=g).cuda())
rp = hip.pack_planes(torch.randn(imgs, 96, H, W, generator=g).cuda())
wp = hip.pack_dense_conv_weight((torch.randn(96, 96, 3, 3, generator=g) / 30).cuda())
al, be = (0.5 + torch.rand(96, generator=g)).cuda(), torch.randn(96, generator=g).cuda()
run = lambda: hip.dense_conv3x3(xp, wp, al, be, rp, True)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"dense conv {imgs} x 96 x {H} x {W}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (stamped build)")
b = (ctypes.c_ulonglong * 32)()
hip.lib().sdf_debug_read_stamps_dense(b)
for g in (0, 1):
    o = b[16 * g:16 * g + 16]
    q = max(o[6], 1)
    clk = o[7] / max(o[8], 1) * 100e6 / 1e9
    print(f"wave {4 * g}: {o[6]} steps; cycles per step: request-next {o[0]/q:.0f}  mfma {o[2]/q:.0f}  store-next {o[5]/q:.0f}  epilogue {o[3]/q:.0f}; "
          f"weight loads {o[4]} cycles in all; kernel {o[7]} cycles = {o[7]/q:.0f} per step, clock {clk:.2f} GHz")
PY
