#!/usr/bin/env python3
"""The plain product on row-major digit planes (csrc/ms_res.hip through sdf_spike_gemm_fwd) a few times - for rocprofv3 passes and
tools/res_ablate.sh.  usage: res_gemm_one.py [M N K]   (default: the third decoder level's stacked-tap product, 17 280 x 864 x 416)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd
M, N, K = [int(v) for v in sys.argv[1:4]] if len(sys.argv) >= 4 else (17280, 864, 416)
dev = "cuda:0"
A = (rnd((M, K), 1) < -0.4).to(torch.uint8).to(dev)
dg = hip.split_weight_i8x3(rnd((N, K), 2, -0.07, 0.07).to(dev))
out = torch.empty((M, N), device=dev)
for _ in range(20):
    hip.spike_gemm(A, dg, out, M, N, K)
torch.cuda.synchronize()
