#!/usr/bin/env python3
"""Eager single-stream forwards of BASELINE config 2 (for the rocprofv3 passes of tools/pmc_forward*.sh, prof_forward_one.sh).
usage: forward_one.py [count] [lif|psn] [R = 1]        R > 1: forward_replicas over R samples (the headline's launch sequence)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
model, sd = bench.build_model(sys.argv[2] if len(sys.argv) > 2 else "lif", dev)
R = int(sys.argv[3]) if len(sys.argv) > 3 else 1
x = torch.cat([bench.synthetic_chunk(1235 + i) for i in range(R)], 0).to(dev)
with torch.no_grad():
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
        model.forward_replicas(x) if R > 1 else model(x)
torch.cuda.synchronize()
