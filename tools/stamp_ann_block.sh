#!/usr/bin/env bash
# diagnostic build of the one-launch attention half block (csrc/ann_block.hip) with in-kernel 100 MHz timestamps: where does a
# workgroup's time go (prologue: x rows + LayerNorm | per head: barrier, weight store + barrier, projections, barrier, attention |
# epilogue)?  The diagnostic library lives beside, not over, the product one.
# usage: tools/stamp_ann_block.sh build      (off the GPU box: build/stamp/libann.so travels with the snapshot)
#        tools/stamp_ann_block.sh [plain|shifted]   (GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p build/stamp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -fno-slp-vectorize -Wno-pass-failed -Wno-unused-function -DSDF_STAMP ${SDF_EXTRA_FLAGS:-} -c sdformerflow_amd/csrc/ann_block.hip -o build/stamp/ann_block.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/stamp/libann.so build/stamp/ann_block.o $(ls sdformerflow_amd/csrc/obj/*.o | grep -v ann_block)
  echo build/stamp/libann.so; exit 0
fi
SDF_HIP_LIB=${SDF_STAMP_LIB:-build/stamp/libann.so} python3 - "${1:-shifted}" <<'PY'
import ctypes, sys, os, torch
sys.path.insert(0, os.getcwd())
from sdformerflow_amd import hip
from sdformerflow_amd.STSwinNet.swin_transformer3D_v2 import SwinTransformerBlock3D
shifted = sys.argv[1] == "shifted"
torch.manual_seed(3)
blk = SwinTransformerBlock3D(96, 3, (2, 9, 9), (1, 4, 4) if shifted else (0, 0, 0), 4.0, True).eval().to("cuda")
x = torch.randn((8, 2, 72, 96, 96), device="cuda")
for _ in range(10):
    blk(x)
torch.cuda.synchronize()
b = (ctypes.c_ulonglong * 72)()
hip.lib().sdf_debug_read_stamps_ann(b)
names = ["entry -> requests issued", "x rows + LayerNorm + weight requests"]
for g in range(3):
    names += [f"head {g}: barrier (previous head done)", f"head {g}: weight store + barrier", f"head {g}: Q K V projections", f"head {g}: barrier (K, V complete)", f"head {g}: attention + projection step"]
names += ["epilogue (shortcut + store)"]
for wg, slot in ((0, 0), (300, 1), (600, 2)):
    o = list(b[24 * slot:24 * slot + 24])
    idx = [0, 1] + [2 + 5 * g + k for g in range(3) for k in range(5)] + [17]
    print(f"workgroup {wg}, wave 0 (100 MHz ticks -> us): total {(o[17] - o[0]) / 100:.2f} us")
    print(f"    prologue: slice-map entry arrived {(o[18] - o[0]) / 100:.2f} | projection weights arrived {(o[19] - o[18]) / 100:.2f} | rows of x arrived {(o[20] - o[19]) / 100:.2f} | "
          f"LayerNorm + split {(o[21] - o[20]) / 100:.2f} | table row + weight requests issued {(o[1] - o[21]) / 100:.2f} us")
    prev = o[0]
    for i, nm in zip(idx[1:], names[1:]):
        print(f"    {nm:45s} {(o[i] - prev) / 100:6.2f} us")
        prev = o[i]
PY
