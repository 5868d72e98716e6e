#!/usr/bin/env python3
"""Every Linear-layer shape of the en4 model at 288x384 (B=1) under the small-tile persistent kernel (SDF_GEMM_WS=0) and the
ping-pong kernel (SDF_GEMM_WS=2): fp32 epilogue (BN + residual) and fused-neuron epilogues.  Run on the GPU box."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
torch.manual_seed(0)
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 2

def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def f32(M, N, K, resid=True):
    A = (torch.rand((M, K), device=dev) < 0.3).to(torch.uint8)
    Wp = hip.split_weight(torch.randn((N, K), device=dev) * 0.05, ns)
    al, be = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1
    out = torch.randn((M, N), device=dev)
    r = {}
    for name, ws, wgs in (("base", "0", "3"), ("b2", "0", "2"), ("b4", "0", "4"), ("b6", "0", "6"), ("pp", "2", "3")):
        os.environ["SDF_GEMM_WS"] = ws
        os.environ["SDF_GEMM_WGS"] = wgs
        r[name] = timeit(lambda: hip.spike_gemm(A, Wp, out, M, N, K, alpha=al, beta=be, resid=out if resid else None))
    print(f"f32  M={M:6d} N={N:4d} K={K:4d} resid={int(resid)}: wg/cu 2|3|4|6 {r['b2']:6.1f} {r['base']:6.1f} {r['b4']:6.1f} {r['b6']:6.1f} us   pp {r['pp']:6.1f} us")

def fused(T, pos, N, K):
    M = pos * T
    A = (torch.rand((M, K), device=dev) < 0.3).to(torch.uint8)
    Wp = hip.split_weight(torch.randn((N, K), device=dev) * 0.05, ns)
    al, be = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1
    out = torch.zeros((M, N), dtype=torch.uint8, device=dev)
    p = hip.NeuronParams("lif", 2.0, 0.1, None)
    r = {}
    for name, ws, wgs in (("base", "0", "3"), ("b2", "0", "2"), ("b4", "0", "4"), ("b6", "0", "6"), ("pp", "2", "3")):
        os.environ["SDF_GEMM_WS"] = ws
        os.environ["SDF_GEMM_WGS"] = wgs
        r[name] = timeit(lambda: hip.spike_gemm_sn(A, Wp, out, N, K, T, pos, pos, 0, pos, p, alpha=al, beta=be))
    print(f"sn{T:<2d} M={M:6d} N={N:4d} K={K:4d}        : wg/cu 2|3|4|6 {r['b2']:6.1f} {r['base']:6.1f} {r['b4']:6.1f} {r['b6']:6.1f} us   pp {r['pp']:6.1f} us")

tok = [69120, 17280, 4320, 1080]
win = [71280, 19440, 6480, 3240]        # window-padded rows of the attention GEMMs (2 x B_ x 81)
for s in range(4):
    C = 96 * 2 ** s
    fused(2, win[s] // 2, 2 * C, C)      # q | k stacked
    f32(win[s], C, C, True)              # proj (+ scatter in the model)
    fused(10, tok[s] // 10, 4 * C, C)    # fc1
    f32(tok[s], C, 4 * C, True)          # fc2
    if s < 3:
        f32(tok[s] // 4, 2 * C, 4 * C, False)   # patch merge
print("decoder tap GEMMs (M = imgs*h*w, N = 9*Cout, K = padded Cin)")
f32(1080, 3456, 1536, False)
f32(4320, 1728, 800, False)
f32(17280, 864, 416, False)
