#!/usr/bin/env python3
"""Run the MS MLP of one swin block a few times (for rocprofv3 --pmc / --kernel-trace).  usage: mlp_one.py B D H W C [three]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd
import mlp_bench
B, D, H, W, Cc = (int(v) for v in sys.argv[1:6]) if len(sys.argv) >= 6 else (1, 10, 72, 96, 96)
three = len(sys.argv) > 6 and sys.argv[6] == "three"
Ch = 4 * Cc
x = rnd((B, D, H, W, Cc), 1, -0.5, 1.0).to("cuda:0")
fc1 = mlp_bench.L(rnd((Ch, Cc), 2, -0.3, 0.3), rnd((Ch,), 3, 0.5, 1.5), rnd((Ch,), 4, -0.2, 0.2), 2)
fc2 = mlp_bench.L(rnd((Cc, Ch), 5, -0.1, 0.1), rnd((Cc,), 6, 0.5, 1.5), rnd((Cc,), 7, -0.2, 0.2), 2)
p = hip.NeuronParams("lif", 2.0, 0.1, None)
for _ in range(8):
    hip.ms_mlp(x, fc1, fc2, p, p, three_launches=three)
torch.cuda.synchronize()
