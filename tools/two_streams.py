#!/usr/bin/env python3
"""Throughput with 1, 2 and 3 forwards in flight on separate HIP streams (one sample each)."""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
model, sd = bench.build_model("lif", torch.device("cuda", 0))
chunk = bench.synthetic_chunk().cuda()
K = 24
with torch.no_grad():
    for _ in range(3): ref = model(chunk)
    torch.cuda.synchronize()
    for ns in (1, 2, 3):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        outs = [None] * ns
        for s in streams:                       # warm the per-stream allocator pools / workspaces
            with torch.cuda.stream(s): model(chunk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            with torch.cuda.stream(streams[i % ns]):
                outs[i % ns] = model(chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(a, b) for o in outs for a, b in zip(o["flow"], ref["flow"]))
        print(f"{ns} stream(s): {K / dt:7.1f} samples/s  ({dt / K * 1e3:.3f} ms per forward), outputs equal to the single-stream run: {same}")

# ---- the same with every in-flight forward captured as a HIP graph (no CPU launch cost)
with torch.no_grad():
    for ns in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        graphs, ins, outs = [], [], []
        for s in streams:
            x = chunk.clone()
            with torch.cuda.stream(s):
                for _ in range(2): model(x)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                o = model(x)
            graphs.append(g); ins.append(x); outs.append(o)
        torch.cuda.synchronize()
        for i in range(ns):
            with torch.cuda.stream(streams[i]): graphs[i].replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            with torch.cuda.stream(streams[i % ns]):
                graphs[i % ns].replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = all(torch.equal(a, b) for o in outs for a, b in zip(o["flow"], ref["flow"]))
        print(f"{ns} graph stream(s): {K / dt:7.1f} samples/s  ({dt / K * 1e3:.3f} ms per forward), outputs equal: {same}")
