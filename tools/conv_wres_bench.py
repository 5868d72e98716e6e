"""A/B timing of the 3x3 spike convolution on the roofline shape (10 x 144 x 192, 96 -> 96): weight-resident kernel vs the
streaming ping-pong kernel; L3-resident (one operand set) and HBM (rotating sets).  usage: python tools/conv_wres_bench.py [ns]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip  # noqa: E402

ns = sys.argv[1] if len(sys.argv) > 1 else "2"
dev = "cuda:0"
imgs, H, W, C = 10, 144, 192, 96
wt = torch.randn((C, C, 3, 3), device=dev) * 0.05
Wp = hip.pack_conv_weight_i8x3(wt) if ns == "i8x3" else hip.pack_conv_weight(wt, int(ns))
al, be = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
sets = [((torch.rand((imgs, H, W, C), device=dev) < 0.3).to(torch.uint8), torch.empty((imgs * H * W, C), device=dev),
         torch.rand((imgs * H * W, C), device=dev), torch.empty((imgs * H * W, C), dtype=torch.uint8, device=dev)) for _ in range(4)]
n = H * W
sn = hip.NeuronParams("lif", 2.0, 0.1, None)


def f32(st):
    hip.spike_conv2d(st[0], Wp, imgs, H, W, C, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=st[1], alpha=al, beta=be, resid=st[2])


def fused(st):
    hip.spike_conv2d(st[0], Wp, imgs, H, W, C, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out_spike=st[3], alpha=al, beta=be, sn=sn, sn_T=10,
                     pos=(n, n, 10 * n, n))


def fused_m(st):
    hip.spike_conv2d(st[0], Wp, imgs, H, W, C, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=st[1], out_spike=st[3], alpha=al, beta=be,
                     resid=st[2], sn=sn, sn_T=10, pos=(n, n, 10 * n, n))


def timed(fn, ss, iters=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(4):
        fn(ss[i % len(ss)])
    e0.record()
    for i in range(iters):
        fn(ss[i % len(ss)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


flop = 2.0 * imgs * H * W * C * 9 * C
for name, fn in (("fp32+resid", f32), ("fused LIF T=10", fused), ("fused LIF + membrane", fused_m)):
    for mode in ("1",) if ns == "i8x3" else ("1", "0"):
        os.environ["SDF_CONV_WRES"] = mode
        hip.reload_switches()
        t_l3, t_hbm = timed(fn, sets[:1]), timed(fn, sets)
        print(f"{name:22s} planes {ns} {'weight-resident' if mode == '1' else 'streaming      '}: L3-resident {t_l3:7.1f} us ({flop / t_l3 / 1e6:6.1f} TF/s)  "
              f"rotating {t_hbm:7.1f} us ({flop / t_hbm / 1e6:6.1f} TF/s)")
os.environ.pop("SDF_CONV_WRES", None)
hip.reload_switches()
