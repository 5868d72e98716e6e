#!/usr/bin/env python3
"""The ANN swin block's Linear layers at BASELINE config 3's sizes: csrc/dense_linear.hip against F.linear (fp32 library GEMM)."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from sdformerflow_amd import hip


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


g = torch.Generator().manual_seed(0)
tot_lib = tot_own = 0.0
for stage, (M, Cc, blocks) in enumerate([(8 * 2 * 72 * 96, 96, 2), (8 * 2 * 36 * 48, 192, 2), (8 * 2 * 18 * 24, 384, 6)]):
    for name, K, N, gelu, res in [("qkv", Cc, 3 * Cc, False, False), ("proj", Cc, Cc, False, True), ("fc1", Cc, 4 * Cc, True, False), ("fc2", 4 * Cc, Cc, False, True)]:
        a = torch.randn(M, K, generator=g).cuda()
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).cuda() if res else None
        wp = hip.pack_dense_linear_weight(w)

        def lib():
            y = F.linear(a, w, b)
            if gelu:
                y = F.gelu(y)
            if res:
                y = r + y
            return y
        t_lib, t_own = timed(lib), timed(lambda: hip.dense_linear(a, wp, b, gelu, r))
        gf = 2 * M * K * N / 1e9
        tot_lib += blocks * t_lib; tot_own += blocks * t_own
        print(f"stage {stage} {name:4s} M={M:6d} K={K:4d} N={N:4d}: library {t_lib:7.1f} us, own {t_own:7.1f} us ({gf / t_own * 1e3:6.1f} TFLOP/s algorithmic), "
              f"max diff {((hip.dense_linear(a, wp, b, gelu, r) - lib()).abs().max() / lib().abs().max()).item():.1e}")
print(f"all Linear layers of one config-3 forward (2 + 2 + 6 blocks): library {tot_lib / 1e3:.2f} ms, own {tot_own / 1e3:.2f} ms")
