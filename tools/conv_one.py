#!/usr/bin/env python3
"""Run one spike-conv shape a few times (for rocprofv3 --pmc / timing).
usage: conv_one.py imgs H W Cin Cout stride [fused|fusedm|resid|plain] [nsplit | i8x3]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
imgs, H, W, Cin, Cout, stride = (int(v) for v in sys.argv[1:7])
mode = sys.argv[7] if len(sys.argv) > 7 else "plain"
fused = mode in ("fused", "fusedm")
ns = sys.argv[8] if len(sys.argv) > 8 else "2"
dev = "cuda:0"
x = (torch.rand((imgs, H, W, Cin), device=dev) < 0.3).to(torch.uint8)
wt = torch.randn((Cout, Cin, 3, 3), device=dev) * 0.05
Wp = hip.pack_conv_weight_i8x3(wt) if ns == "i8x3" else hip.pack_conv_weight(wt, int(ns))
ns = 1.5 if ns == "i8x3" else int(ns)
al, be = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
out = torch.empty((imgs * OH * OW, Cout), device=dev)
outs = torch.empty((imgs * OH * OW, Cout), dtype=torch.uint8, device=dev)
res = torch.randn((imgs * OH * OW, Cout), device=dev) if mode in ("resid", "fusedm") else None
def run():
    if fused:
        n = OH * OW
        hip.spike_conv2d(x, Wp, imgs, H, W, Cin, OH, OW, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=out if mode == "fusedm" else None,
                         out_spike=outs, alpha=al, beta=be, resid=res, sn=hip.NeuronParams("lif", 2.0, 0.1, None), sn_T=10,
                         pos=(n // 10 * 10 if False else n, n, 10 * n, n) if imgs == 10 else (n, n, 0, n))
    else:
        hip.spike_conv2d(x, Wp, imgs, H, W, Cin, OH, OW, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=out, alpha=al, beta=be, resid=res)
for _ in range(3): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 5 * 1e3
fl = 2.0 * imgs * OH * OW * Cout * 9 * Cin
print(f"conv imgs={imgs} {H}x{W} Cin={Cin} Cout={Cout} s={stride} {mode} nsplit={ns}: {us:.1f} us  {fl/us/1e6:.1f} TF (algorithmic), x{ns} planes = {ns*fl/us/1e6:.1f} TF")
