#!/usr/bin/env python3
"""Headline configuration (en4, 10 bins, 288 x 384) with B samples per forward and F forwards in flight (HIP-graph replay):
samples/s for every (B, F) asked.   usage: batch_try.py [lif|psn] B:F [B:F ...]     e.g.  batch_try.py lif 1:3 3:1 3:2 6:1"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
from sdformerflow_amd.harness import prepare_chunk
from sdformerflow_amd.synthetic import synth_voxel
kind = sys.argv[1]
dev = torch.device("cuda:0")
model, _ = bench.build_model(kind, dev)
rep = os.environ.get("SDF_REPLICAS", "0") == "1"
for spec in sys.argv[2:]:
    B, F_ = (int(v) for v in spec.split(":"))
    chunks = [torch.cat([prepare_chunk(synth_voxel(1, 10, 288, 384, seed=1235 + j * B + b)) for b in range(B)], 0) for j in range(F_)]
    if rep:
        model.replicas = True
    r = bench.inflight_rate(model, chunks, dev, F_, max(30, 240 // B))
    print(f"{kind} B={B} F={F_}: {B * r['samples_per_s']:.1f} samples/s, {r['ms_per_step']:.3f} ms per forward of {B}, latency {r['latency_ms_single_stream']:.3f} ms", flush=True)
    torch.cuda.empty_cache()
