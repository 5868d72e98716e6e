#!/usr/bin/env python3
"""Dense 3x3 convolution (csrc/dense_conv_wres.hip) against the library's fp32 convolution on BASELINE config 3's patch-embedding
shape: 16 images (batch 8 x 2 temporal chunks) x 96 channels x 288 x 384.  usage: dense_conv_bench.py [imgs H W]"""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from sdformerflow_amd import hip

imgs, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 288, 384)
g = torch.Generator().manual_seed(0)
x = torch.randn(imgs, 96, H, W, generator=g).cuda()
w = (torch.randn(96, 96, 3, 3, generator=g) / 30).cuda()
al, be = (0.5 + torch.rand(96, generator=g)).cuda(), torch.randn(96, generator=g).cuda()
xp, wp = hip.pack_planes(x), hip.pack_dense_conv_weight(w)
rp = hip.pack_planes(torch.randn(imgs, 96, H, W, generator=g).cuda())


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gf = imgs * H * W * 96 * 864 * 2 / 1e9
t_lib = timed(lambda: F.conv2d(x, w, None, 1, 1))
t_own = timed(lambda: hip.dense_conv3x3(xp, wp, al, be, None, True))
t_res = timed(lambda: hip.dense_conv3x3(xp, wp, al, be, rp, True))
t_f32 = timed(lambda: hip.dense_conv3x3(xp, wp, al, be, rp, True, True))
t_pack = timed(lambda: hip.pack_planes(x))
y = hip.unpack_planes(hip.dense_conv3x3(xp, wp), 96)
ref = F.conv2d(x, w, None, 1, 1)
print(f"{imgs} x 96 x {H} x {W}: {gf:.1f} GFLOP; library fp32 conv {t_lib:.3f} ms; "
      f"own conv+BN+ReLU {t_own:.3f} ms ({gf / t_own:.0f} TFLOP/s algorithmic, {3 * gf / t_own:.0f} on the fp16 pipe), "
      f"+resid {t_res:.3f} ms, +resid fp32 out {t_f32:.3f} ms, pack {t_pack:.3f} ms; "
      f"max |own - library| {((y - ref).abs().max() / ref.abs().max()).item():.2e} of max")
