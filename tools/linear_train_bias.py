#!/usr/bin/env python3
"""Is sdf_linear_train_fwd's dX product BIASED?  Signed error statistics against fp64 on a large random problem, beside the library's
fp32 product.  A gradient that is a sum of dX over 10^7 cancelling terms (a PSN bias) sees a one-sided error of 1e-8 of an element."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
torch.manual_seed(0)
for M, N, K in ((200000, 96, 96), (50000, 384, 96), (50000, 96, 384)):
    dy = torch.randn((M, N), device=dev) * 1e-3
    w = torch.randn((N, K), device=dev) * 0.05
    ref = dy.double() @ w.double()
    for name, got in (("ours", hip.linear_train(dy, w, mode=1)), ("library", dy @ w)):
        e = got.double() - ref
        toward0 = float((e * torch.sign(ref)).mean() / ref.abs().mean())          # > 0: magnitudes too large, < 0: shrunk
        up = float(e.mean() / ref.abs().mean())                                   # signed: toward +inf / -inf
        rms = float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        print(f"dX {M}x{N}x{K} {name:8s}: rms error {rms:.2e} of rms; mean signed error {up:+.2e} of mean |dx|; mean error along sign(dx) {toward0:+.2e}")
# the same statistics for dW = dY^T X (sdf_linear_dw_fwd): every element is a 276 480-long sum
for M, N, K in ((276480, 96, 96), (276480, 96, 384)):
    dy = torch.randn((M, N), device=dev) * 1e-3
    x = (torch.rand((M, K), device=dev) < 0.2).float()
    ref = dy.double().t() @ x.double()
    for name, got in (("ours", hip.linear_dw(dy, x)), ("library", dy.t() @ x)):
        e = got.double() - ref
        print(f"dW {M}x{N}x{K} {name:8s}: rms error {float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()):.2e} of rms; mean signed error "
              f"{float(e.mean() / ref.abs().mean()):+.2e} of mean |dw|; mean error along sign(dw) {float((e * torch.sign(ref)).mean() / ref.abs().mean()):+.2e}")
