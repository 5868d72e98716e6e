#!/usr/bin/env bash
# diagnostic build of the ping-pong kernel with in-kernel cycle stamps (workgroup 0): where does the time go, per role?
# usage (on the GPU box): tools/stamp_pp.sh imgs H W Cin Cout stride nsplit [fused|resid]
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -DSDF_STAMP -c sdformerflow_amd/csrc/spike_mm_pp.hip -o /tmp/pp_stamp.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libsdf_stamp.so /tmp/pp_stamp.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v spike_mm_pp)
# the diagnostic library lives beside, not over, the product one
SDF_HIP_LIB=/tmp/libsdf_stamp.so python3 - "$@" <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
from sdformerflow_amd import hip
imgs, H, W, Cin, Cout, stride, nsplit = (int(v) for v in sys.argv[1:8])
mode = sys.argv[8] if len(sys.argv) > 8 else ""
dev = "cuda:0"
x = (torch.rand((imgs, H, W, Cin), device=dev) < 0.3).to(torch.uint8)
Wp = hip.pack_conv_weight(torch.randn((Cout, Cin, 3, 3), device=dev) * 0.05, nsplit)
al, be = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
M = imgs * OH * OW
out = torch.empty((M, Cout), device=dev)
outs = torch.empty((M, Cout), dtype=torch.uint8, device=dev)
res = torch.randn((M, Cout), device=dev) if mode == "resid" else None
def run():
    if mode == "fused":
        n = OH * OW
        hip.spike_conv2d(x, Wp, imgs, H, W, Cin, OH, OW, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out_spike=outs, alpha=al, beta=be,
                         sn=hip.NeuronParams("lif", 2.0, 0.1, None), sn_T=10, pos=(n, n, 0, n))
    else:
        hip.spike_conv2d(x, Wp, imgs, H, W, Cin, OH, OW, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=out, alpha=al, beta=be, resid=res)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print(f"conv {imgs}x{H}x{W} {Cin}->{Cout} s={stride} nsplit={nsplit} {mode}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us (stamped build)")
b = (ctypes.c_ulonglong * 16)()
hip.lib().sdf_debug_read_stamps_pp(b)
Q = max(b[4], 1)
print(f"producer wave 8 (wg 0): Q={Q} stages; per stage: wait-empty {b[0]/Q:.0f}  store+signal {b[1]/Q:.0f}  decode {b[2]/Q:.0f}  load-issue {b[3]/Q:.0f} cycles")
for g, o in ((0, 5), (1, 10)):
    nt = max(b[o + 4], 1)
    print(f"consumer group {g} wave 0: tiles {nt}; per tile: wait-full {b[o]/nt:.0f}  mfma {b[o+1]/nt:.0f}  epilogue {b[o+2]/nt:.0f}; total {b[o+3]} cycles")
PY
