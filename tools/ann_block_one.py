#!/usr/bin/env python3
"""Time the attention half of BASELINE config 3's first-stage block (B = 8, D = 2, 72 x 96 tokens, C = 96, 704 windows of 162):
the one-launch kernel (csrc/ann_block.hip) against the four launches it replaces (LayerNorm, qkv Linear, window attention, proj Linear).
usage: ann_block_one.py [plain|shifted] [fused|four] [mlp]     (mlp: time the MLP half instead - one launch vs LayerNorm + fc1 + fc2)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
from sdformerflow_amd.STSwinNet.swin_transformer3D_v2 import SwinTransformerBlock3D
shifted = len(sys.argv) > 1 and sys.argv[1] == "shifted"
which = sys.argv[2] if len(sys.argv) > 2 else "fused"
mlp_half = len(sys.argv) > 3 and sys.argv[3] == "mlp"
if which == "four":
    os.environ["SDF_ANN_BLOCK"] = os.environ["SDF_ANN_MLP"] = "0"
dev = "cuda:0"
torch.manual_seed(3)
blk = SwinTransformerBlock3D(96, 3, (2, 9, 9), (1, 4, 4) if shifted else (0, 0, 0), 4.0, True).eval().to(dev)
x = torch.randn((8, 2, 72, 96, 96), device=dev)
class _Shortcut(torch.nn.Module):              # the attention half only: the MLP half returns its shortcut
    def forward(self, y, resid=None):
        return resid
if mlp_half:                                    # the MLP half only: the attention half returns its shortcut
    blk.attn.forward_rows = lambda y2, row_map, B_, mask, resid=None, norm=None: resid
else:
    blk.mlp = _Shortcut()
    os.environ["SDF_ANN_MLP"] = "0"
hip.reload_switches()                      # (the switches are read once: say that the environment changed)
for _ in range(5): blk(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps): blk(x)
e1.record(); torch.cuda.synchronize()
if mlp_half:
    print(f"MLP half block, {which}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (includes the LayerNorm(norm1) launch of the attention half when that is not fused)")
else:
    print(f"attention half block, {'shifted' if shifted else 'plain'} windows, {which}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (includes the LayerNorm(norm2) launch of the MLP half)")
