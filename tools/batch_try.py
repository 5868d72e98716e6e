#!/usr/bin/env python3
"""Headline configuration (en4, 10 bins, 288 x 384) in bench.py's scheme: F HIP streams, each replaying the graph of one launch sequence
over R independent samples (model.forward_replicas): samples/s for every (R, F) asked, flows verified bit-equal to plain forwards.
usage: batch_try.py [lif|psn] R:F[:steps] ...     e.g.  batch_try.py lif 1:3 4:2 6:2:960"""
import os, sys, torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import bench
kind = sys.argv[1]
dev = torch.device("cuda:0")
model, _ = bench.build_model(kind, dev)
for spec in sys.argv[2:]:
    R, F_, *st = (int(v) for v in spec.split(":"))
    steps = st[0] if st else 96
    r = bench.inflight_rate(model, dev, F_, R, max(steps, R * F_) // (R * F_) * (R * F_))
    print(f"{kind} R={R} F={F_}: {r['samples_per_s']:.1f} samples/s ({r['steps']} samples), {r['ms_per_step']:.3f} ms per sample, latency {r['latency_ms_single_stream']:.3f} ms", flush=True)
    torch.cuda.empty_cache()
