#!/usr/bin/env python3
"""Does the whole forward capture into a HIP graph (torch.cuda.CUDAGraph), and what does a replay cost?"""
import os, sys, time, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import bench
model, sd = bench.build_model("lif", torch.device("cuda", 0))
chunk = bench.synthetic_chunk().cuda()
with torch.no_grad():
    for _ in range(3): out = model(chunk)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): out = model(chunk)
    torch.cuda.synchronize()
    print("eager  ms/step", (time.perf_counter() - t0) / 20 * 1e3)
    # CPU-side cost of one forward's launches (no sync inside)
    t0 = time.perf_counter(); out = model(chunk); t1 = time.perf_counter(); torch.cuda.synchronize()
    print("CPU launch time of one forward (queue empty at start) ms", (t1 - t0) * 1e3)
    static_in = chunk.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): model(static_in)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        static_out = model(static_in)
    torch.cuda.synchronize()
    ref = [f.clone() for f in model(static_in)["flow"]]
    g.replay(); torch.cuda.synchronize()
    print("graph replay equals eager:", all(torch.equal(a, b) for a, b in zip(ref, static_out["flow"])))
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print("graph  ms/step", (time.perf_counter() - t0) / 20 * 1e3)
