#!/usr/bin/env python3
"""Small-M / large-K convolutions of the U-Net tail under split-K variants (SDF_KSPLIT_MULT).  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def case(imgs, H, W, Cin, Cout, KH=3, KW=3):
    x = (torch.rand((imgs, H, W, Cin), device=dev) < 0.3).to(torch.uint8)
    w = torch.randn((Cout, Cin, KH, KW), device=dev) * 0.05
    Wp = hip.pack_conv_weight(w, 2)
    al, be = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    out = torch.empty((imgs * H * W, Cout), device=dev)
    dy = (-1, 0, 1)[:KH] if KH == 3 else (0, 1)[:KH]
    dx = (-1, 0, 1)[:KW] if KW == 3 else (0, 1)[:KW]
    r = []
    for m in ("1", "2", "3", "4"):
        os.environ["SDF_KSPLIT_MULT"] = m
        hip.reload_switches()
        r.append(timeit(lambda: hip.spike_conv2d(x, Wp, imgs, H, W, Cin, H, W, KH, KW, 1, dy, dx, out=out, alpha=al, beta=be)))
    fl = 2.0 * imgs * H * W * Cout * KH * KW * Cin
    print(f"conv {imgs}x{H}x{W} {Cin}->{Cout} {KH}x{KW}: ksplit x1|x2|x3|x4 " + " ".join(f"{v:6.1f}" for v in r) + f" us   ({fl / min(r) / 1e6:.0f} TF best)")
case(10, 9, 12, 768, 768)                 # U-Net res-block
for KH, KW in ((1, 1), (1, 2), (2, 1), (2, 2)):
    case(10, 9, 12, 1536, 384, KH, KW)    # decoder 0 parity classes
for KH, KW in ((1, 1), (2, 2)):
    case(10, 18, 24, 784, 192, KH, KW)    # decoder 1
    case(10, 36, 48, 400, 96, KH, KW)     # decoder 2
