"""Margin-explained parity of the FREE-RUNNING GPU forward (tests/replay.py): every neuron layer's spikes of the actual
forward are delta-consistent with the reference neuron on the pre-activation the oracle computes from the GPU's own upstream
spikes - 0 unexplained decisions - and the oracle replay ends in the GPU's flow maps to fp32 tolerance.  Replaces the
"mismatch rate <= x" reading of the stage tests with: every departure from the reference's spikes sits within delta of the
reference's own threshold, nothing else differs beyond floating-point rounding."""
import os

import numpy as np
import pytest
import torch
import yaml

import replay
from oracle import sdformer_oracle as O
from sdformerflow_amd import harness
from sdformerflow_amd.loss.flow_supervised import AEE
from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet, MS_SpikingformerFlowNet_en4
from sdformerflow_amd.synthetic import synth_label, synth_state_dict, synth_voxel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
CFG = os.path.join(HERE, "..", "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")
FLOW_TOL = 2e-5          # max |gpu flow - replayed flow| / max |flow|: fp32 accumulation-order noise through the last linear layers


def build(kind, size=(288, 384), en4=True, T=10, bins=10, window=(2, 9, 9), psn_bias=None):
    cfg = yaml.safe_load(open(CFG))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind, num_steps=T)
    cfg["model"]["num_bins"] = bins
    cfg["swin_transformer"].update(input_size=list(size), window_size=list(window))
    if not en4:
        cfg["swin_transformer"].update(swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    cls = MS_SpikingformerFlowNet_en4 if en4 else MS_SpikingformerFlowNet
    model = cls(cfg["model"].copy(), cfg["swin_transformer"].copy())
    sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd, strict=True)
    ocfg = {"neuron": O.NeuronCfg(kind, 0.1, None, 2.0, T), "num_bins": bins, "window_size": window,
            "depths": cfg["swin_transformer"]["swin_depths"], "num_heads": cfg["swin_transformer"]["swin_num_heads"]}
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    return model.eval().to(DEV), sd, ocfg


def check(kind, B, size, seed, en4=True, planes=2, **kw):
    model, sd, ocfg = build(kind, size, en4, **kw)
    model.gemm_nsplit = planes
    bins = ocfg["num_bins"]
    chunk = harness.prepare_chunk(synth_voxel(B, bins, size[0], size[1], seed=seed))
    flows, ref, report = replay.run(model.engine(), chunk.to(DEV), chunk, sd, lambda c: O.forward_flownet(c, sd, ocfg))
    summ = replay.summarise(report)
    worst = sorted((r for r in report if r["forced"]), key=lambda r: -r["needed_ulps"])[:4]
    print(f"{kind} B={B} {size} planes={planes}: {summ}")
    for r in worst:
        print(f"    {r['layer']:90s} flips {r['flips']:7d} ambiguous {r['ambiguous']:8d} of {r['n']:10d}  needed {r['needed_ulps']:.1f} ulps")
    assert summ["layers_free"] <= summ["layers_forced"] // 4, summ     # only the integer-input token gates (and dead attn_sn) run unforced
    assert summ["unexplained"] == 0, [r for r in report if r["forced"] and r["unexplained"]][:5]
    assert summ["ambiguous"] <= 2e-5 * summ["decisions"], summ           # delta is tight: it covers a tiny part of the decisions
    devs = []
    for g, r in zip(flows, ref):
        g = g.cpu()
        assert g.shape == r.shape and torch.isfinite(g).all()
        devs.append(float((g - r).abs().max() / r.abs().max()))
    print(f"    flows vs replayed reference: max-abs-dev / max|flow| per scale {['%.1e' % d for d in devs]}")
    assert max(devs) <= FLOW_TOL, devs
    # the plain forward (no tape: one-launch kernels write nothing but their outputs, the decoders' skip inputs go through one
    # multi-descriptor neuron launch) is bit-equal to the taped one that was just replayed
    plain = model(chunk.to(DEV))["flow"]
    assert all(torch.equal(a, b) for a, b in zip(plain, flows)), "the untaped forward differs from the taped one"
    return flows, ref, summ


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_config2_free_running_forward_is_a_delta_consistent_execution_of_the_reference(kind):
    """BASELINE configs[1] at full size (1 x 10 x 2 x 288 x 384, en4), the default 2-plane weights."""
    flows, ref, _ = check(kind, 1, (288, 384), 1235)
    # north star "AEE within 1e-3 of reference", on the same forward: (a) AEE of the GPU flow with the replayed reference flow
    # as the label - the distance between the two, in pixels - against the flow's own magnitude; (b) AEE of both against a
    # random label through the product's metric class
    label, mask = synth_label(1, 288, 384)
    g, r = flows[-1].cpu(), ref[-1]
    dist = float(AEE(g, r, torch.ones_like(mask), 1.0)()[0][0])
    assert dist <= 1e-3 * float(r.abs().mean()), (dist, float(r.abs().mean()))
    a_g, a_r = float(AEE(g, label, mask, 1.0)()[0][0]), float(AEE(r, label, mask, 1.0)()[0][0])
    print(f"    AEE(gpu, replay as label) {dist:.2e} px (mean |flow| {float(r.abs().mean()):.3f}); vs random label: {a_g:.6f} / {a_r:.6f}")
    assert abs(a_g - a_r) <= 1e-3 * a_r


def test_exact_three_plane_weights():
    check("lif", 1, (288, 384), 1236, planes=3)


def test_batch_of_two_three_encoders():
    check("lif", 2, (144, 192), 77, en4=False)
    check("psn", 2, (144, 192), 78, en4=False)


def test_plif_and_slttlif_models_three_encoders():
    """The neuron types no shipped configuration uses (reference Spiking_modules.py:49-56, 75-82) through the whole fused engine:
    plif's multiplicative charge (k = sigmoid(w), w = -0.1 from the synthetic state) reaches every fused kernel through `tau`."""
    check("plif", 1, (144, 192), 79, en4=False)
    check("SLTTlif", 1, (144, 192), 80, en4=False)
    with pytest.raises(Exception, match="no fused kernel"):
        build("glif", (144, 192), en4=False)[0].engine()


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_log_true_attention_scores_equal_the_replayed_oracle(kind):
    """`model(x, log=True)["attn"]` (reference Spiking_STSwinNet.py:283-284): the attention score `attn_sn(gated k)` of the last block
    of every stage, (T', B_, Wh, Ww, C).  In the spike-forced replay the oracle's gated tensor IS the GPU's, and `attn_sn` reads 0 / 1
    inputs (no rounding anywhere): the scores must be bit-equal.  (The reference's own `get_layer_attention_scores` raises for every
    model - oracle/sdformer_oracle.py `swin_encoder`; the score itself is pinned on the reference module's second return value in
    tests/test_oracle_golden.py.)"""
    size = (144, 192)
    model, sd, ocfg = build(kind, size, en4=False)
    chunk = harness.prepare_chunk(synth_voxel(1, 10, size[0], size[1], seed=4242))
    want = []
    flows, ref, report = replay.run(model.engine(), chunk.to(DEV), chunk, sd, lambda c: O.forward_flownet(c, sd, ocfg, want))
    assert replay.summarise(report)["unexplained"] == 0
    out = model(chunk.to(DEV), log=True)
    assert all(torch.equal(a, b) for a, b in zip(out["flow"], flows))                     # logging does not disturb the forward
    assert len(out["attn"]) == len(want) == 3
    for s, (got, w) in enumerate(zip(out["attn"], want)):
        C = 96 << s
        assert got.shape[0] == 2 and tuple(got.shape[2:]) == (9, 9, C) and got.dtype == torch.float32
        assert torch.equal(got.cpu().reshape(w.shape), w), f"stage {s}"
        assert 0.005 < float(w.mean()) < 0.995, float(w.mean())
    assert model(chunk.to(DEV))["attn"] is None


def test_config5_shape_T20_odd_sizes():
    """20 bins / T = 20, a width whose stage maps need padding and cropping (88 -> 90, 44 -> 45, 22 -> 27, 11 -> 18; reduced from
    480 x 640 so that the oracle finishes in seconds; the smallest stage must still hold one 9 x 9 window, as in the reference)."""
    check("lif", 1, (288, 352), 91, T=20, bins=20)


def test_mdr_configuration_window8_T5_psn():
    check("psn", 1, (256, 256), 92, en4=False, T=5, window=(2, 8, 8))


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_sew_family_free_running_forward(kind):
    """`SpikingformerFlowNet` (SEW shortcuts: the stream carries sums of spikes; dense library GEMMs / convolutions where a
    layer reads it, spike kernels where a layer reads spikes, the fused SEW window-attention kernel with the shift mask) -
    the same statement as for the MS models: every neuron layer delta-consistent, flows equal to the replayed reference."""
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import SpikingformerFlowNet
    cfg = yaml.safe_load(open(CFG))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind)
    cfg["swin_transformer"].update(input_size=[144, 192], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    model = SpikingformerFlowNet(cfg["model"].copy(), cfg["swin_transformer"].copy())
    sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()
                           if not k.endswith(("relative_position_index",))})
    model.load_state_dict(sd, strict=False)
    model = model.eval().to(DEV)
    sd = {k: v for k, v in model.state_dict().items()}
    sd = {k: v.cpu() for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    ocfg = {"neuron": O.NeuronCfg(kind, 0.1, None, 2.0, 10), "num_bins": 10, "window_size": (2, 9, 9), "depths": [2, 2, 6],
            "num_heads": [3, 6, 12]}
    chunk = harness.prepare_chunk(synth_voxel(1, 10, 144, 192, seed=1234 + 7))
    flows, ref, report = replay.run(model.engine(), chunk.to(DEV), chunk, sd, lambda c: O.forward_sew_flownet(c, sd, ocfg))
    summ = replay.summarise(report)
    print(f"SEW {kind}: {summ}")
    assert summ["layers_forced"] == 75 and summ["layers_free"] == 0 and summ["unexplained"] == 0, summ
    devs = [float((g.cpu() - r).abs().max() / r.abs().max()) for g, r in zip(flows, ref)]
    print("    flows vs replayed reference:", ["%.1e" % d for d in devs])
    assert max(devs) <= FLOW_TOL, devs
    out = model(chunk.to(DEV))
    assert out["attn"] is None and all(torch.equal(a, b) for a, b in zip(out["flow"], flows))


def test_odd_feature_sizes_three_encoders():
    """150 x 200: stage maps 38 x 50, 19 x 25, 10 x 13 - the transposed-convolution decoders overshoot their skips by one pixel and
    `skip_concat` (models/model_util.py:14-19) crops them; padded windows at every stage."""
    check("lif", 1, (150, 200), 93, en4=False)
