#!/usr/bin/env python3
"""Forward and dX of the spike-fed Linear layers of a configs[3] training step (local batch 4): sdf_linear_train_fwd against the
library products it replaces (torch: rocBLAS fp32).  usage (GPU box): python3 tools/linear_train_bench.py"""
import os, sys, torch
import torch.nn.functional as F
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
tot = [0.0] * 4
for stage, (M, C, blocks) in enumerate([(276480, 96, 2), (69120, 192, 2), (17280, 384, 6), (4320, 768, 2)]):
    for name, N, K, per in (("q/k/proj", C, C, 3), ("fc1", 4 * C, C, 1), ("fc2", C, 4 * C, 1)):
        x = (torch.rand((M, K), device=dev) < 0.2).float()
        dy = torch.randn((M, N), device=dev) * 1e-3
        w = torch.randn((N, K), device=dev) * 0.05
        f0, f1 = t(lambda: hip.linear_train(x, w, None, 0)), t(lambda: F.linear(x, w))
        b0, b1 = t(lambda: hip.linear_train(dy, w, None, 1)), t(lambda: dy @ w)
        for i, v in enumerate((f0, f1, b0, b1)): tot[i] += v * per * blocks
        print(f"stage {stage} {name:9s} M={M:6d} N={N:4d} K={K:4d}: forward ours {f0:7.1f} us  library {f1:7.1f} us | dX ours {b0:7.1f} us  library {b1:7.1f} us")
        del x, dy, w
print(f"all of a step's swin blocks: forward ours {tot[0] / 1e3:.2f} ms, library {tot[1] / 1e3:.2f} ms; dX ours {tot[2] / 1e3:.2f} ms, library {tot[3] / 1e3:.2f} ms")
