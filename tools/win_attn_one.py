#!/usr/bin/env python3
"""Time the fused window-attention kernel (sdf_win_attn_fwd) on BASELINE config 3's stage shapes.
usage: win_attn_one.py [ann|sew] [B_] [nH] [N] [mask|nomask]  (default: ann 704 3 162 mask = STTFlowNet stage 0, batch 8,
shifted block)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
mode = sys.argv[1] if len(sys.argv) > 1 else "ann"
B_ = int(sys.argv[2]) if len(sys.argv) > 2 else 704
nH = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N = int(sys.argv[4]) if len(sys.argv) > 4 else 162
dev, C = "cuda:0", 32 * nH
nW = 88 if B_ % 88 == 0 else 1
g = torch.Generator(device=dev).manual_seed(5)
bias = torch.randn((nH, N, N), device=dev, generator=g)
mask = (torch.rand((nW, N, N), device=dev, generator=g) < 0.2).float() * -100.0
if len(sys.argv) > 5 and sys.argv[5] == "nomask":
    mask = None
if mode == "ann":
    qkv = torch.randn((B_, N, 3 * C), device=dev, generator=g)
    scale = torch.full((nH,), 10.0, device=dev)
    run = lambda: hip.win_attn_ann(qkv, scale, bias, mask, nH)
else:
    q, k, v = ((torch.rand((2, B_, N // 2, C), device=dev, generator=g) < 0.3).to(torch.uint8) for _ in range(3))
    scale = torch.full((nH,), 32 ** -0.5, device=dev)
    run = lambda: hip.win_attn_sew(q, k, v, scale, bias, mask, nH, 2, B_, N // 2)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
fl = 4.0 * N * N * 32 * B_ * nH                         # QK^T + P.V, SURVEY.md 8(d): 3.36 MFLOP per (window, head)
NT = (N + 15) // 16
f16 = N % 2 == 0 and NT in (8, 11) and os.environ.get("SDF_ATTN_F32") != "1" and os.environ.get("SDF_ATTN_GENERIC") != "1"
if f16 and mode == "ann":
    # three v_mfma_f32_16x16x32_f16 (16 cycles) per S tile + six v_mfma_f32_16x16x16_f16 (8 cycles) per P.V tile pair
    cyc = B_ * nH * NT * NT * (3 * 16 + 6 * 8)
    what = "hi/lo fp16 planes, three products: 9 MFMAs per tile pair on the 16-bit pipe"
elif f16:
    cyc = B_ * nH * NT * NT * (1 * 16 + 4 * 8)           # binary q / k / v: one exact product for S, P = hi + lo against binary v
    what = "binary operands exact in fp16: 5 MFMAs per tile pair on the 16-bit pipe"
else:
    cyc = B_ * nH * NT * NT * 16 * 32                  # v_mfma_f32_16x16x4_f32 (32 cycles): 8 per S tile + 8 per P.V tile pair
    what = "16 fp32 MFMAs per tile pair"
floor_us = cyc / 4 / 256 / 2400.0                       # 4 SIMDs x 256 CUs, 2.4 GHz
print(f"win_attn {mode} B_={B_} nH={nH} N={N} {'mask' if mask is not None else 'no mask'}: {us:.1f} us  {fl/us/1e6:.1f} TFLOP/s algorithmic = {100*fl/us/1e6/157.3:.1f} % of the "
      f"157.3 TF fp32-MFMA peak; {what} -> pipe floor {floor_us:.1f} us, MFMA-pipe utilisation {100*floor_us/us:.1f} %")
