#!/usr/bin/env python3
"""One shape of sdf_linear_dw_fwd a few times (for rocprofv3 --pmc / --kernel-trace).
usage: linear_dw_one.py M N K            (Linear form)
       linear_dw_one.py conv imgs C H W   (3x3 convolution form, N = C)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
if sys.argv[1] == "conv":
    imgs, Cc, H, W = (int(v) for v in sys.argv[2:6])
    x = (torch.rand((imgs, Cc, H, W), device=dev) < 0.2).float()
    dy = torch.randn((imgs, Cc, H, W), device=dev) * 1e-3
    run = lambda: hip.conv3x3_dw(dy, x)
else:
    M, N, K = (int(v) for v in sys.argv[1:4])
    dy = torch.randn((M, N), device=dev) * 1e-3
    x = (torch.rand((M, K), device=dev) < 0.2).float()
    run = lambda: hip.linear_dw(dy, x)
for _ in range(8): run()
torch.cuda.synchronize()
