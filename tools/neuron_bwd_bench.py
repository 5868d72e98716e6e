#!/usr/bin/env python3
"""HBM rate of the neuron backward kernels on the stage-0 hidden tensor of config 2 (T = 10, 69 120 x 384 neurons).
Algorithmic bytes per neuron-step: x 4 B + dL/ds 4 B in, dL/dx 4 B out."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
T, N = 10, 6912 * 384
dev = "cuda:0"
x = torch.rand((T, N), device=dev) * 0.6 - 0.3
g = torch.randn((T, N), device=dev)
W, b = torch.randn((T, T), device=dev) * 0.3, torch.full((T,), -0.1, device=dev)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, fn in (("lif_bwd", lambda: hip.lif_bwd(x, g, 2.0, 0.1, None, True, 2.0)),
                 ("psn_bwd (dx, dW, db)", lambda: hip.psn_bwd(x, W, b, g, 2.0)),
                 ("lif_fwd (fp32 spikes)", lambda: hip.lif_fwd(x, 2.0, 0.1, None, torch.float32))):
    us = timeit(fn)
    byts = T * N * (12 if "bwd" in name else 8)
    print(f"{name:24s} T={T} N={N}: {us:8.1f} us  {byts/us/1e6:6.2f} TB/s algorithmic = {100*byts/us/1e6/8.0:.0f} % of 8 TB/s")
