#!/usr/bin/env python3
"""Tile-configuration sweep (SDF_GEMM_CFG 0..3) of the small-tile spike GEMM on the attention layers' shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
torch.manual_seed(0)
os.environ["SDF_GEMM_WS"] = "0"
def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def sweep(label, fn):
    r = []
    for c in "0123":
        os.environ["SDF_GEMM_CFG"] = c
        try:
            r.append(f"{timeit(fn):6.1f}")
        except Exception as e:
            r.append("   n/a")
    os.environ.pop("SDF_GEMM_CFG")
    print(f"{label}: cfg0 512x96 | cfg1 256x96 | cfg2 256x32 | cfg3 128x32 = " + " ".join(r) + f"   default {timeit(fn):6.1f} us")
win = [71280, 19440, 6480, 3240]
tok = [69120, 17280, 4320, 1080]
for s in range(4):
    C = 96 * 2 ** s
    M = win[s]
    A = (torch.rand((M, C), device=dev) < 0.3).to(torch.uint8)
    Wqk = hip.split_weight(torch.randn((2 * C, C), device=dev) * 0.05, 2)
    Wp = hip.split_weight(torch.randn((C, C), device=dev) * 0.05, 2)
    al2, be2 = torch.rand(2 * C, device=dev) + 0.5, torch.randn(2 * C, device=dev) * 0.1
    al, be = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    out2 = torch.zeros((M, 2 * C), dtype=torch.uint8, device=dev)
    out = torch.randn((M, C), device=dev)
    p = hip.NeuronParams("lif", 2.0, 0.1, None)
    sweep(f"q|k  sn2 M={M:6d} N={2*C:4d} K={C:4d}", lambda: hip.spike_gemm_sn(A, Wqk, out2, 2 * C, C, 2, M // 2, M // 2, 0, M // 2, p, alpha=al2, beta=be2))
    sweep(f"proj f32 M={M:6d} N={C:4d} K={C:4d}", lambda: hip.spike_gemm(A, Wp, out, M, C, C, alpha=al, beta=be, resid=out))
    M2 = tok[s]
    A4 = (torch.rand((M2, 4 * C), device=dev) < 0.3).to(torch.uint8)
    W2 = hip.split_weight(torch.randn((C, 4 * C), device=dev) * 0.05, 2)
    o2 = torch.randn((M2, C), device=dev)
    sweep(f"fc2  f32 M={M2:6d} N={C:4d} K={4*C:4d}", lambda: hip.spike_gemm(A4, W2, o2, M2, C, 4 * C, alpha=al, beta=be, resid=o2))
