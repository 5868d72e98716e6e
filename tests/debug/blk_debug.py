#!/usr/bin/env python3
"""Bisect a swin-block mismatch: attention with / without the fused q|k GEMM, MLP with / without the ping-pong kernel."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_engine_gpu as T
model, sd, ocfg = T.build("lif")
eng = model.cuda().engine()
torch.manual_seed(1)
x0 = (torch.randn((1, 10, 72, 96, 96), device="cuda") * 0.5).contiguous()
blk = eng.stages[0][0]
def run_attn(fused):
    saved = blk.qk
    if not fused: blk.qk = None
    y = eng.attention(x0.clone(), blk)
    blk.qk = saved
    return y
a, b = run_attn(True), run_attn(False)
d = (a - b).abs()
print("attention fused-qk vs separate: mismatch", float((d > 1e-4).float().mean()), "max", float(d.max()))
for ws in ("0", ""):
    if ws: os.environ["SDF_GEMM_WS"] = ws
    else: os.environ.pop("SDF_GEMM_WS", None)
    y = eng.mlp(x0.clone(), blk)
    if ws: base = y
print("mlp pp vs base: mismatch", float(((y - base).abs() > 1e-4).float().mean()), "max", float((y - base).abs().max()))

# ---- isolate: fused q|k GEMM halves vs the separate GEMMs, then the strided gate vs the contiguous one
from sdformerflow_amd import hip
from sdformerflow_amd.STSwinNet_SNN.Spiking_swin_transformer3D import get_window_size
B, D, H, W, Cc = x0.shape
ws, ss = get_window_size((D, H, W), blk.window_size, blk.shift_size)
rowmap, B_ = eng._slice_map(B, D, H, W, ws, ss)
Tq, N1 = ws[0], ws[1] * ws[2]
rows = B_ * N1
n, M = rows * Cc, Tq * rows
xs = torch.empty((M, Cc), dtype=torch.uint8, device="cuda")
hip.neuron_fwd(x0, xs, Tq, 1, n, 0, 0, 0, n, blk.sn_proj, rowmap=rowmap, rowlen=Cc)
q = torch.empty((M, Cc), dtype=torch.uint8, device="cuda"); k = torch.empty_like(q)
hip.spike_gemm_sn(xs, blk.q.Wp, q, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_q, alpha=blk.q.alpha, beta=blk.q.beta)
hip.spike_gemm_sn(xs, blk.k.Wp, k, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_k, alpha=blk.k.alpha, beta=blk.k.beta, add=blk.pe, add_prows=N1)
qk = torch.empty((M, 2 * Cc), dtype=torch.uint8, device="cuda")
hip.spike_gemm_sn(xs, blk.qk["Wp"], qk, 2 * Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_q, alpha=blk.qk["alpha"], beta=blk.qk["beta"], add=blk.qk["add"], add_prows=N1)
print("pe shape", tuple(blk.pe.shape), "q half mismatch", float((qk[:, :Cc] != q).float().mean()), "k half mismatch", float((qk[:, Cc:] != k).float().mean()),
      "rates", float(q.float().mean()), float(k.float().mean()))
e1 = torch.empty_like(q); e2 = torch.empty_like(q)
hip.qk_gate(q, k, e1, Tq, rows, Cc, blk.sn2_q)
qk2 = torch.cat([q, k], 1).contiguous()
hip.qk_gate(qk2, qk2[:, Cc:], e2, Tq, rows, Cc, blk.sn2_q, ldq=2 * Cc, ldk=2 * Cc)
print("gate strided vs contiguous mismatch", float((e1 != e2).float().mean()), "rate", float(e1.float().mean()))
