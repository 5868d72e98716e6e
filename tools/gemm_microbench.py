#!/usr/bin/env python3
"""Time the spike GEMM variants on the shapes of the en4 forward (HIP events on the launch stream)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
CFGS = ("ws", "1", "2", "3")   # tile configs (see sdf_spike_gemm_fwd); "auto" = library heuristic

def timeit(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def run(tag, M, N, K, T=0, HW=None):
    A = (torch.rand((M, K), device=dev) < 0.3).to(torch.uint8)
    W = torch.randn((N, K), device=dev) * 0.1
    Wp = hip.split_weight(W, 3)
    al, be = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1
    res = []
    for cfg in CFGS:
        os.environ["SDF_GEMM_WS"] = "1" if cfg == "ws" else "0"
        os.environ["SDF_GEMM_CFG"] = cfg
        if T == 0:
            out = torch.empty((M, N), device=dev)
            us = timeit(lambda: hip.spike_gemm(A, Wp, out, M, N, K, alpha=al, beta=be))
        else:
            out = torch.empty((M, N), dtype=torch.uint8, device=dev)
            pos = M // T
            p = hip.NeuronParams("lif", 2.0, 0.1, None)
            if HW is None:
                us = timeit(lambda: hip.spike_gemm_sn(A, Wp, out, N, K, T, pos, pos, 0, pos, p, alpha=al, beta=be))
            else:
                us = timeit(lambda: hip.spike_gemm_sn(A, Wp, out, N, K, T, pos, HW, T * HW, HW, p, alpha=al, beta=be))
        res.append(us)
    fl = 2.0 * M * N * K
    print(f"{tag:18s} M={M:6d} N={N:5d} K={K:5d} T={T:2d} " + " ".join(f"{c}:{r:6.1f}us" for c, r in zip(CFGS, res)) + f"  best {fl/min(res)/1e6:6.1f} TF")

for s, (C, hw, rows) in enumerate([(96, 72 * 96, 440 * 81), (192, 36 * 48, 120 * 81), (384, 18 * 24, 30 * 81), (768, 9 * 12, 10 * 81)]):
    run(f"s{s} q/k fused T=2", 2 * rows, C, C, 2)
    run(f"s{s} proj f32", 2 * rows, C, C)
    run(f"s{s} fc1 fused T=10", 10 * hw, 4 * C, C, 10, hw)
    run(f"s{s} fc1 f32", 10 * hw, 4 * C, C)
    run(f"s{s} fc2 f32", 10 * hw, C, 4 * C)
