#!/usr/bin/env python3
"""Run one spike-GEMM shape a few times (for rocprofv3 --pmc).  usage: gemm_one.py M N K T cfg [planes = 2]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
M, N, K, T = (int(v) for v in sys.argv[1:5])
os.environ["SDF_GEMM_CFG"] = sys.argv[5]
hip.reload_switches()
dev = "cuda:0"
A = (torch.rand((M, K), device=dev) < 0.3).to(torch.uint8)
Wp = hip.split_weight(torch.randn((N, K), device=dev) * 0.1, int(sys.argv[6]) if len(sys.argv) > 6 else 2)
al, be = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1
for _ in range(8):
    if T == 0:
        out = torch.empty((M, N), device=dev)
        hip.spike_gemm(A, Wp, out, M, N, K, alpha=al, beta=be)
    else:
        out = torch.empty((M, N), dtype=torch.uint8, device=dev)
        pos = M // T
        hip.spike_gemm_sn(A, Wp, out, N, K, T, pos, pos, 0, pos, hip.NeuronParams("lif", 2.0, 0.1, None), alpha=al, beta=be)
torch.cuda.synchronize()
