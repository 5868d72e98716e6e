#!/usr/bin/env bash
# diagnostic build of the weight-resident conv kernel with in-kernel cycle stamps (workgroup 0, wave 0 of each group): where
# does a step go?  The diagnostic library lives beside, not over, the product one.  usage (GPU box): tools/stamp_wres.sh [ns] [f32|fused|fusedm]
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_STAMP -c sdformerflow_amd/csrc/spike_conv_wres.hip -o /tmp/wres_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsdf_stamp.so /tmp/wres_stamp.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v spike_conv_wres)
SDF_HIP_LIB=/tmp/libsdf_stamp.so SDF_CONV_WRES=2 python3 - "$@" <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
from sdformerflow_amd import hip
ns = sys.argv[1] if len(sys.argv) > 1 else "2"
mode = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = "cuda:0"
imgs, H, W, C = 10, 144, 192, 96
x = (torch.rand((imgs, H, W, C), device=dev) < 0.3).to(torch.uint8)
wt = torch.randn((C, C, 3, 3), device=dev) * 0.05
Wp = hip.pack_conv_weight_i8x3(wt) if ns == "i8x3" else hip.pack_conv_weight(wt, int(ns))
al, be = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
out = torch.empty((imgs * H * W, C), device=dev)
res = torch.rand((imgs * H * W, C), device=dev)
sp = torch.empty((imgs * H * W, C), dtype=torch.uint8, device=dev)
n = H * W
sn = hip.NeuronParams("lif", 2.0, 0.1, None)
def run():
    if mode == "f32":
        hip.spike_conv2d(x, Wp, imgs, H, W, C, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=out, alpha=al, beta=be, resid=res)
    else:
        hip.spike_conv2d(x, Wp, imgs, H, W, C, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=out if mode == "fusedm" else None, out_spike=sp,
                         alpha=al, beta=be, resid=res if mode == "fusedm" else None, sn=sn, sn_T=10, pos=(n, n, 10 * n, n))
for _ in range(20): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
print(f"{mode} planes {ns}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us per launch (stamped build)")
b = (ctypes.c_ulonglong * 32)()
hip.lib().sdf_debug_read_stamps_wres(b)
for g in (0, 1):
    o = b[16 * g:16 * g + 16]
    q = max(o[6], 1)
    clk = o[7] / max(o[8], 1) * 100e6 / 1e9
    print(f"group {g} wave 0: {o[6]} steps; per step: issue-next {o[0]/q:.0f}  wait-halo {o[1]/q:.0f}  mfma {o[2]/q:.0f}  epilogue {o[3]/q:.0f}  hand-over {o[4]/q:.0f};"
          f" weight load {o[5]} ; kernel {o[7]} cycles, clock {clk:.2f} GHz")
PY
