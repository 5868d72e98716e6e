#!/usr/bin/env python3
"""Eager single-stream forwards of BASELINE config 2 (for the rocprofv3 passes of tools/pmc_forward*.sh, prof_forward_one.sh).
usage: forward_one.py [count] [lif|psn]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
model, sd = bench.build_model(sys.argv[2] if len(sys.argv) > 2 else "lif", dev)
x = bench.synthetic_chunk().to(dev)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        model(x)
torch.cuda.synchronize()
