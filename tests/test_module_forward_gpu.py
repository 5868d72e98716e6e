"""`forward()` of the block-level classes (the reference's own entry points below the model: Spiking_swin_transformer3D.py
:661, :164-181, :824-847, :952-974 and the SEW siblings) against the outputs of the REAL reference's modules on the same
seeded weights and inputs (fixtures qk_attention / ms_block / sew_attention, tests/golden/make_golden.py) and, for the SEW
classes the fixtures do not hold, against the oracle.  An element counts as touched by a spike flip when it is off by more
than 1e-4 of the tensor's mean magnitude; the rest must agree to 2e-5 (tests/test_replay_gpu.py is where flips are explained)."""
import os

import numpy as np
import pytest
import torch

from oracle import sdformer_oracle as O
from sdformerflow_amd.STSwinNet_SNN import Spiking_swin_transformer3D as SW
from sdformerflow_amd.synthetic import synth_state_dict, synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def kw(kind, T):
    return {"num_steps": T, "v_reset": None, "v_th": 0.1, "neuron_type": kind, "surrogate_fun": "surrogate.ATan()", "tau": 2.0,
            "detach_reset": True, "spike_norm": "BN"}


def load_synth(mod):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if not k.endswith(("relative_position_index",))}
    sd = synth_state_dict(shapes)
    mod.load_state_dict(sd, strict=False)
    return mod.to(DEV).eval(), {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}


def close(got, ref, max_rate):
    got, ref = got.float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = ref.abs().mean().item() + 1e-12
    d = (got - ref).abs()
    bad = d > 1e-4 * scale
    rate = bad.float().mean().item()
    rest = d[~bad].max().item() / scale if (~bad).any() else 0.0
    assert rate <= max_rate and rest <= 2e-5, (rate, rest)
    return rate


@pytest.mark.parametrize("tag,kind", [("c96_lif", "lif"), ("c96_psn", "psn"), ("c192_lif", "lif"), ("c384_psn", "psn")])
def test_qk_window_attention_module_matches_the_reference_module(tag, kind):
    g = np.load(os.path.join(G, "qk_attention.npz"))
    B_, C, nH, seed = (int(v) for v in g[f"{tag}_cfg"])
    m, _ = load_synth(SW.Spiking_QK_WindowAttention3D(C, (2, 9, 9), (0, 0, 0), nH, norm="BN", **kw(kind, 10)))
    y, attn = m(rnd((2, B_, 9, 9, C), seed, -0.5, 1.0).to(DEV))
    print(tag, "flip-touched", close(y, g[f"{tag}_y"], 2e-3))
    # second return value: `attn_sn` on the gated tensor (reference :709-711; what log=True collects) against the reference module's.
    # Binary in, binary out: it differs only where an upstream q / k / token spike flipped at its threshold (the same rate as above)
    gs = np.load(os.path.join(G, "qk_attention_scores.npz"))
    want = np.unpackbits(gs[f"{tag}_attn"])[:attn.numel()].reshape(tuple(attn.shape))
    assert attn.dtype == torch.float32 and tuple(attn.shape) == (2, B_, 9, 9, C)
    miss = float((attn.cpu().numpy() != want).mean())
    print(tag, "attention-score mismatches", miss)
    assert miss <= 2e-3 and abs(float(attn.mean()) - float(gs[f"{tag}_rate"])) <= 2e-3
    if kind == "lif":                                                          # return_attention=True of the block (:836-837)
        blk, _ = load_synth(SW.MS_Spiking_SwinTransformerBlock3D(C, (18, 18), nH, window_size=(2, 9, 9), shift_size=(0, 0, 0),
                                                                 norm_layer="BN", **kw(kind, 4)))
        sc = blk(rnd((1, 4, 18, 18, C), 19, -0.5, 1.0).to(DEV), None, return_attention=True)
        assert tuple(sc.shape) == (2, 8, 9, 9, C) and 0.01 < float(sc.mean()) < 0.99


@pytest.mark.parametrize("tag,kind", [("lif_sw", "lif"), ("lif_w", "lif"), ("psn_sw", "psn")])
def test_ms_block_module_matches_the_reference_module(tag, kind):
    g = np.load(os.path.join(G, "ms_block.npz"))
    H, W, *shift = (int(v) for v in g[f"{tag}_cfg"])
    blk, _ = load_synth(SW.MS_Spiking_SwinTransformerBlock3D(96, (H, W), 3, window_size=(2, 9, 9), shift_size=tuple(shift),
                                                             norm_layer="BN", **kw(kind, 4)))
    x = rnd((1, 4, H, W, 96), 17, -0.5, 1.0).to(DEV)
    x0 = x.clone()
    y = blk(x, None)
    assert torch.equal(x, x0)                                            # the caller's tensor is not updated in place
    print(tag, "flip-touched", close(y, g[f"{tag}_y"], 5e-3))


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_ms_patch_merging_and_mlp_modules(kind):
    g = np.load(os.path.join(G, "ms_block.npz"))
    pm, _ = load_synth(SW.MS_SpikingPatchMerging((9, 21), 96, norm_layer="BN", **kw(kind, 4)))
    close(pm(rnd((1, 4, 9, 21, 96), 19, -0.5, 1.0).to(DEV)), g[f"{kind}_merge_y"], 1e-3)
    mlp, sd = load_synth(SW.MS_Spiking_Mlp(96, 384, norm_layer="BN", **kw(kind, 4)))
    x = rnd((4, 2, 9, 12, 96), 31, -0.5, 1.0)
    with torch.no_grad():
        ref = O.ms_mlp(x, sd, "", O.NeuronCfg(kind, 0.1, None, 2.0, 4))
    close(mlp(x.to(DEV)), ref, 2e-3)


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_sew_block_mlp_and_merging_modules_against_the_oracle(kind):
    n = O.NeuronCfg(kind, 0.1, None, 2.0, 4)
    blk, sd = load_synth(SW.Spiking_SwinTransformerBlock3D(96, (18, 21), 3, window_size=(2, 9, 9), shift_size=(1, 4, 4), qk_scale=0.125,
                                                           norm_layer="BN", **kw(kind, 4)))
    x = (rnd((1, 4, 18, 21, 96), 41) > 0.5).float() + (rnd((1, 4, 18, 21, 96), 42) > 0.7).float()     # a sum of spike tensors
    with torch.no_grad():
        ref = O.sew_block(x, sd, "", 3, (2, 9, 9), (1, 4, 4), n)
    close(blk(x.to(DEV), None), ref, 1e-2)                                # outputs are small integers: a flip moves a whole element
    mlp, sd = load_synth(SW.Spiking_Mlp(96, 384, norm_layer="BN", **kw(kind, 4)))
    xm = rnd((4, 2, 9, 12, 96), 43, 0.0, 2.0)
    with torch.no_grad():
        ref = O.sew_mlp(xm, sd, "", n)
    close(mlp(xm.to(DEV)), ref, 5e-3)
    pm, sd = load_synth(SW.SpikingPatchMerging((9, 21), 96, norm_layer="BN", **kw(kind, 4)))
    xp = rnd((1, 4, 9, 21, 96), 44, 0.0, 2.0)
    with torch.no_grad():
        ref = O.sew_patch_merge(xp, sd, "", n)
    close(pm(xp.to(DEV)), ref, 5e-3)


@pytest.mark.parametrize("kind,C,H,W", [("lif", 96, 18, 24), ("psn", 96, 9, 12), ("lif", 768, 9, 12)])
def test_ms_resblock_module_against_the_oracle(kind, C, H, W):
    """`MS_ResBlock.forward` (reference Spiking_modules.py:906-933) at module level; C = 768 at 9 x 12 is the U-Net bottleneck: the
    small-M kernel with both fused epilogues."""
    from sdformerflow_amd.STSwinNet_SNN import Spiking_modules as SM
    rb, sd = load_synth(SM.MS_ResBlock(C, C, 1, "ADD", **kw(kind, 10)))
    x = rnd((10, 1, C, H, W), 41, -0.5, 1.0)
    with torch.no_grad():
        ref = O.ms_resblock(x, sd, "", O.NeuronCfg(kind, 0.1, None, 2.0, 10))
    xd = x.to(DEV)
    y = rb(xd)
    assert torch.equal(xd.cpu(), x) and tuple(y.shape) == tuple(x.shape)
    close(y, ref, 5e-3)


def test_module_forward_refuses_cpu_tensors():
    from sdformerflow_amd.hip import SdfError
    m = SW.MS_SpikingPatchMerging((9, 21), 96, norm_layer="BN", **kw("lif", 4)).eval()
    with pytest.raises(SdfError):
        m(torch.zeros(1, 4, 9, 21, 96))
