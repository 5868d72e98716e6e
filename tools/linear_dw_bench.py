#!/usr/bin/env python3
"""dW = dY^T X of the spike-fed Linear layers of a configs[3] training step (local batch 4): sdf_linear_dw_fwd against the library
product it replaces (torch: rocBLAS fp32).  usage (GPU box): python3 tools/linear_dw_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
def t(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
tot_a = tot_b = 0.0
for stage, (M, C, blocks) in enumerate([(276480, 96, 2), (69120, 192, 2), (17280, 384, 6), (4320, 768, 2)]):
    for name, N, K, per in (("q/k/proj", C, C, 3), ("fc1", 4 * C, C, 1), ("fc2", C, 4 * C, 1)):
        dy = torch.randn((M, N), device=dev) * 1e-3
        x = (torch.rand((M, K), device=dev) < 0.2).float()
        ours = t(lambda: hip.linear_dw(dy, x))
        libt = t(lambda: dy.t() @ x)
        err = float((hip.linear_dw(dy, x) - dy.t() @ x).abs().max() / (dy.t() @ x).abs().max())
        tot_a += ours * per * blocks; tot_b += libt * per * blocks
        print(f"stage {stage} {name:9s} M={M:6d} N={N:4d} K={K:4d}: ours {ours:7.1f} us  library {libt:7.1f} us  ({4.0 * M * (N + K) / ours / 1e6:5.2f} TB/s, "
              f"{2.0 * M * N * K / ours / 1e6:6.1f} TF)  max diff / max {err:.1e}")
        del dy, x
print(f"all dW of a step's swin blocks: ours {tot_a / 1e3:.2f} ms, library {tot_b / 1e3:.2f} ms")
# the convolution form: MS_ResBlock convolutions of the patch embedding (local batch 4 x 10 steps = 40 images of 144 x 192) and of the bottleneck
for imgs, Cc, H, W in ((40, 96, 144, 192), (40, 768, 9, 12)):
    x = (torch.rand((imgs, Cc, H, W), device=dev) < 0.2).float()
    dy = torch.randn((imgs, Cc, H, W), device=dev) * 1e-3
    w = torch.randn((Cc, Cc, 3, 3), device=dev)
    ours = t(lambda: hip.conv3x3_dw(dy, x), 5)
    rows = t(lambda: (hip._ringed_rows(dy), hip._ringed_rows(x)), 5)
    libt = t(lambda: torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=1, padding=1), 5)
    a, b = hip.conv3x3_dw(dy, x), torch.nn.grad.conv2d_weight(x, w.shape, dy, stride=1, padding=1)
    print(f"conv dW imgs={imgs} C={Cc} {H}x{W}: ours {ours:7.1f} us (of which the two ringed channels-last copies {rows:7.1f})  library {libt:7.1f} us  "
          f"max diff / max {float((a - b).abs().max() / b.abs().max()):.1e}")
    del x, dy
