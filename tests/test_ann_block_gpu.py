"""The attention half of an ANN video-swin block as one launch (csrc/ann_block.hip: LayerNorm -> qkv -> cosine window attention ->
proj -> + x through the window slice map) against the CPU oracle's `ann_block` (oracle/sdformer_oracle.py, itself pinned to the
reference's swin_transformer3D_v2.py:272-336 by tests/test_oracle_golden.py) and against the four-launch path it replaces."""
import os

import pytest
import torch

from oracle import sdformer_oracle as O
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _block(shift, qkv_bias, seed):
    from sdformerflow_amd.STSwinNet.swin_transformer3D_v2 import SwinTransformerBlock3D
    blk = SwinTransformerBlock3D(96, 3, (2, 9, 9), shift, 4.0, qkv_bias).eval()
    sd = {}
    for i, (k, v) in enumerate(blk.state_dict().items()):
        if k.endswith(("relative_position_index", "relative_coords_table")):
            continue
        if k.endswith("logit_scale"):
            sd[k] = rnd(tuple(v.shape), seed + i, 1.0, 4.8)                  # exp -> 2.7 ... 100 (clamped at ln 100 = 4.6)
        elif "norm" in k:
            sd[k] = rnd(tuple(v.shape), seed + i, 0.5, 1.5) if k.endswith("weight") else rnd(tuple(v.shape), seed + i, -0.3, 0.3)
        elif k.endswith("bias"):
            sd[k] = rnd(tuple(v.shape), seed + i, -0.2, 0.2)
        else:
            sd[k] = rnd(tuple(v.shape), seed + i, -0.15, 0.15)
    blk.load_state_dict(sd, strict=False)
    return blk, sd


@pytest.mark.parametrize("qkv_bias", [True, False])
@pytest.mark.parametrize("shift", [(0, 0, 0), (1, 4, 4)])
@pytest.mark.parametrize("B,D,H,W", [(2, 2, 18, 27), (1, 2, 20, 30), (3, 4, 9, 9)])
def test_block_equals_oracle_and_the_four_launch_path(B, D, H, W, shift, qkv_bias):
    """Both halves of the block as one launch each (csrc/ann_block.hip, csrc/ann_mlp_block.hip).  Aligned (18 x 27), padded (20 x 30 -> 27 x 36: padding tokens are zero rows behind the norm, their outputs dropped) and
    single-window-per-slab feature maps; plain and shifted windows (mask, roll); with and without the qkv bias.  Whole block
    (attention half + MLP half) within 3e-5 of the fp32 CPU oracle relative to the output's magnitude; the one-launch half block
    within 2e-5 of the four-launch path (both carry 22-bit products)."""
    blk, sd = _block(shift, qkv_bias, 400 + H)
    x = rnd((B, D, H, W, 96), 77 + W, -1.5, 1.5)
    ref = O.ann_block(x, sd, "", 3, (2, 9, 9), shift)
    blk = blk.to(DEV)
    xg = x.to(DEV)
    assert hip.ann_attn_block_supported(96, 3, 162)
    got = blk(xg).cpu()
    with hip.scoped_switches(SDF_ANN_BLOCK="0", SDF_ANN_MLP="0"):     # LayerNorm, qkv, attention, proj | LayerNorm, fc1 + GELU, fc2: seven launches
        old = blk(xg).cpu()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 3e-5 * scale
    assert (got - old).abs().max().item() <= 2e-5 * scale
    assert torch.equal(xg.cpu(), x)                                       # the input is not written


def test_half_block_alone_against_fp64():
    """x + proj(attention(LN(x))) of one aligned feature map against the same arithmetic in fp64 (the reference's op sequence)."""
    blk, sd = _block((1, 4, 4), True, 500)
    B, D, H, W, C = 2, 2, 18, 18, 96
    x = rnd((B, D, H, W, C), 501, -2.0, 2.0)
    sd64 = {k: v.double() for k, v in sd.items()}
    # fp64: the oracle's block with an identity MLP (fc2 = 0 -> the second half adds nothing)
    sd64["mlp.fc2.weight"] = torch.zeros_like(sd64["mlp.fc2.weight"])
    sd64["mlp.fc2.bias"] = torch.zeros_like(sd64["mlp.fc2.bias"])
    tab, msk = O.relative_coords_table, O.compute_mask
    O.relative_coords_table = lambda *a, **k: tab(*a, **k).double()
    O.compute_mask = lambda *a, **k: msk(*a, **k).double()
    try:
        ref = O.ann_block(x.double(), sd64, "", 3, (2, 9, 9), (1, 4, 4))
    finally:
        O.relative_coords_table, O.compute_mask = tab, msk
    blk = blk.to(DEV)
    with torch.no_grad():
        blk.mlp.fc2.weight.zero_()
        blk.mlp.fc2.bias.zero_()
    got = blk(x.to(DEV)).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_tables_and_weight_planes_follow_the_parameters():
    """The (bias + mask) * log2 e table and the fp16 weight planes are cached per parameter version: an in-place update of the logit
    scale, the cpb_mlp and the qkv / proj weights must reach the next forward."""
    blk, sd = _block((1, 4, 4), True, 600)
    x = rnd((1, 2, 18, 18, 96), 601, -1.5, 1.5)
    blk = blk.to(DEV)
    first = blk(x.to(DEV)).cpu()
    with torch.no_grad():
        blk.attn.logit_scale.add_(0.4)
        blk.attn.cpb_mlp[2].weight.mul_(1.5)
        blk.attn.qkv.weight.mul_(0.9)
        blk.attn.proj.weight.add_(0.01)
    sd2 = {k: v.detach().cpu().clone() for k, v in blk.state_dict().items() if not k.endswith(("relative_position_index", "relative_coords_table"))}
    ref = O.ann_block(x, sd2, "", 3, (2, 9, 9), (1, 4, 4))
    got = blk(x.to(DEV)).cpu()
    assert (got - ref).abs().max().item() <= 3e-5 * ref.abs().max().item()
    assert (got - first).abs().max().item() > 1e-3                          # (the update did change the result)
