"""GPU parity of the whole forward path against the CPU oracle at BASELINE config 1/2 size
(1 x 10 x 2 x 288 x 384, MS_SpikingformerFlowNet_en4, lif and psn).

The random-weight spiking net is chaotic (one flipped spike decorrelates everything downstream, shown
CPU-vs-CPU in DESIGN.md), so parity is *teacher-forced*: every stage of the HIP engine is fed the
oracle's input for that stage, and inside the stage every neuron layer is checked by the spike-forced
replay of tests/replay.py (0 decisions that the reference's own threshold margin does not explain; fp32
outputs to 2e-5).  tests/test_replay_gpu.py makes the same statement for the free-running forward.
"""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import sdformer_oracle as O
from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet_en4
from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel, synth_label

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = os.path.join(os.path.dirname(__file__), "..", "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")


def build(kind, H=288, W=384, cls=MS_SpikingformerFlowNet_en4, T=10):
    cfg = yaml.safe_load(open(CFG))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind, num_steps=T)
    cfg["model"]["num_bins"] = T
    cfg["swin_transformer"]["input_size"] = [H, W]
    if cls.num_en == 3:
        cfg["swin_transformer"].update(swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    model = cls(cfg["model"].copy(), cfg["swin_transformer"].copy())
    sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd, strict=True)
    model.eval()
    ocfg = {"neuron": O.NeuronCfg(kind, 0.1, None, 2.0, T), "num_bins": T, "window_size": (2, 9, 9),
            "depths": [2, 2, 6, 2][:cls.num_en], "num_heads": [3, 6, 12, 24][:cls.num_en]}
    return model, sd, ocfg


def unet_tail_oracle(blocks, sd, n):
    """The oracle's U-Net tail on given encoder features (same code path as O.forward_flownet)."""
    u = "sttmultires_unet."
    y = blocks[-1]
    i = 0
    while u + f"resblocks.{i}.conv1.0.weight" in sd:
        y = O.ms_resblock(y, sd, u + f"resblocks.{i}.", n)
        i += 1
    preds, E = [], len(blocks)
    for i in range(E):
        y = O.skip_concat_ch(y, blocks[E - i - 1])
        if i > 0:
            y = O.skip_concat_ch(preds[-1], y)
        d = u + f"decoders.{i}."
        s = O.neuron(y, n, sd, d + "sn.spiking_neuron.")
        T, B = s.shape[:2]
        z = torch.nn.functional.conv_transpose2d(s.flatten(0, 1), sd[d + "deconv.0.weight"], None, stride=2, padding=1, output_padding=1)
        y = O.bn_ch2(z.view(T, B, *z.shape[1:]), sd, d + "norm_layer.norm_layer.")
        q = u + f"preds.{i}."
        s = O.neuron(y, n, sd, q + "sn.spiking_neuron.")
        preds.append(O._conv_seq(s, sd[q + "conv.0.weight"], sd[q + "conv.0.bias"], padding=0))
    return preds


def stage_replay(name, eng, gpu_call, oracle_call, report, tol=2e-5):
    """One stage, teacher-forced on the oracle's input, checked layer by layer (tests/replay.py): every neuron layer inside
    the stage is delta-consistent with the reference on the GPU's own upstream spikes - 0 unexplained decisions - and the
    stage's fp32 output equals the replayed reference to `tol` of its largest magnitude.  (Round 1 bounded the RATE of
    elements touched by a flip here: 2e-2 for the patch embedding, 1e-1 for the U-Net predictions.)"""
    import replay
    got, ref, rep = replay.run_part(eng, gpu_call, oracle_call)
    summ = replay.summarise(rep)
    got = got if isinstance(got, (list, tuple)) else [got]
    ref = ref if isinstance(ref, (list, tuple)) else [ref]
    dev = max(float((g.float().cpu() - r).abs().max() / r.abs().max()) for g, r in zip(got, ref))
    report.append((name, summ, dev))
    assert summ["unexplained"] == 0, (name, summ, [r for r in rep if r["forced"] and r["unexplained"]][:3])
    assert dev <= tol, (name, dev)
    return ref


def teacher_forced_all(tag, eng, sd, ocfg, chunk, tail=True, stages=None):
    """Every stage of the engine on the oracle's input for that stage, each checked layer by layer (stage_replay)."""
    n, ws = ocfg["neuron"], tuple(ocfg["window_size"])
    shift = tuple(w // 2 for w in ws)
    p = "sttmultires_unet.encoders.swin3d."
    report = []
    # patch embedding: 7 convolutions + 6 neuron layers, every layer checked
    ref = stage_replay("patch_embed", eng, lambda: eng.patch_embed(chunk.to(DEV)).permute(1, 0, 4, 2, 3),
                       lambda: O.patch_embed(chunk, sd, p + "patch_embed.", n, ocfg["num_bins"]), report)[0]
    y = ref.permute(1, 0, 3, 4, 2).contiguous()
    feats, E = [], len(ocfg["depths"])
    for s, (depth, nH) in enumerate(zip(ocfg["depths"], ocfg["num_heads"])):
        if stages is not None and s >= stages:
            break
        for i in range(depth):
            y = stage_replay(f"stage{s}.block{i}", eng, lambda: eng.swin_block(y.contiguous().to(DEV), s, i),
                             lambda: O.ms_block(y, sd, p + f"layers.{s}.swin_blocks.{i}.", nH, ws, (0, 0, 0) if i % 2 == 0 else shift, n),
                             report)[0]
        feats.append(y.contiguous())
        if s < E - 1:
            y = stage_replay(f"stage{s}.merge", eng, lambda: eng.patch_merge(y.contiguous().to(DEV), s),
                             lambda: O.ms_patch_merge(y, sd, p + f"layers.{s}.downsample.", n), report)[0]
    if tail and stages is None:
        # U-Net tail on the oracle's encoder features; compared on its per-scale predictions
        O_feats = [f.permute(1, 0, 4, 2, 3).contiguous() for f in feats]
        stage_replay("unet_tail", eng, lambda: [pp.permute(1, 0, 4, 2, 3) for pp in eng.unet_tail([f.contiguous().to(DEV) for f in feats])],
                     lambda: unet_tail_oracle(O_feats, sd, n), report)
    for name, summ, dev in report:
        print(f"{tag:5s} {name:16s} layers {summ['layers_forced']:2d} flips {summ['flips']:4d} ambiguous {summ['ambiguous']:6d} of "
              f"{summ['decisions']:10d} unexplained {summ['unexplained']} needed {summ['needed_ulps_max']:.1f} ulps; output max-dev {dev:.1e}")
    return report


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_teacher_forced_stage_parity(kind):
    model, sd, ocfg = build(kind)
    chunk = O.prepare_chunk(synth_voxel(1, 10, 288, 384, seed=1235))
    teacher_forced_all(kind, model.to(DEV).engine(), sd, ocfg, chunk)


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_free_running_forward_statistics(kind):
    """Free-running end-to-end: shapes, finiteness, and flow / AEE statistics next to the oracle."""
    model, sd, ocfg = build(kind)
    chunk = O.prepare_chunk(synth_voxel(1, 10, 288, 384, seed=1235))
    with torch.no_grad():
        ref = O.forward_flownet(chunk, sd, ocfg)
    out = model.to(DEV)(chunk.to(DEV))
    assert out["attn"] is None and len(out["flow"]) == 4
    label, mask = synth_label(1, 288, 384)
    aee_ref = float(O.aee(ref[-1], label, mask, 1.0)[0][0])
    aee_got = float(O.aee(out["flow"][-1].cpu(), label, mask, 1.0)[0][0])
    for i, (g, r) in enumerate(zip(out["flow"], ref)):
        g = g.cpu()
        assert g.shape == r.shape == (1, 2, 288, 384) and torch.isfinite(g).all()
        rel = (g - r).abs().mean().item() / r.abs().mean().item()
        print(f"{kind} flow{i}: mean|ref| {r.abs().mean():.3f} mean|got| {g.abs().mean():.3f} mean-abs-dev/mean|ref| {rel:.3e}")
        assert abs(g.abs().mean().item() - r.abs().mean().item()) < 0.15 * r.abs().mean().item()
    print(f"{kind} AEE oracle {aee_ref:.5f} hip {aee_got:.5f} rel {abs(aee_got - aee_ref) / aee_ref:.2e}")
    # a STATISTIC of two decorrelated flows against a random label (the label noise dominates it): kept as a sanity band only.
    # The north star's "AEE within 1e-3 of reference" is asserted where it can be exact: tests/test_replay_gpu.py, on the same
    # free-running forward, against the reference replayed on the GPU's own spikes
    assert abs(aee_got - aee_ref) < 2e-3 * aee_ref


def test_cpu_input_is_refused():
    from sdformerflow_amd import hip
    model, _, _ = build("lif")
    with pytest.raises(hip.SdfError):
        model(torch.zeros(1, 10, 2, 288, 384))


# ---------------------------------------------------------------- module-level drop-ins of a9 / a10 vs the fixtures
def test_ann_window_attention_module_matches_reference_fixture():
    from sdformerflow_amd.STSwinNet.swin_transformer3D_v2 import WindowAttention3D, compute_mask
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ann_attention.npz"))
    m = WindowAttention3D(96, (2, 9, 9), (0, 0, 0), 3, qkv_bias=True)
    assert np.allclose(m.relative_coords_table.numpy(), g["coords_table"], atol=1e-6)
    assert int(m.relative_position_index.sum()) == int(g["rel_index_sum"])
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if "relative" not in k})
    m.load_state_dict(sd, strict=False)
    m = m.eval().to(DEV)
    x = rnd((4, 162, 96), 13, -1.0, 1.0).to(DEV)
    mask = compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4), DEV)
    y, _ = m(x, mask)
    assert np.abs(y.cpu().numpy() - g["y"]).max() <= 5e-5            # reference WindowAttention3D output, fp32
    y2, _ = m(x, None)
    assert np.abs(y2.cpu().numpy() - g["y_nomask"]).max() <= 5e-5


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_sew_window_attention_module_matches_reference_fixture(kind):
    from sdformerflow_amd.STSwinNet.swin_transformer3D_v2 import compute_mask
    from sdformerflow_amd.STSwinNet_SNN.Spiking_swin_transformer3D import Spiking_BN_WindowAttention3D
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sew_attention.npz"))
    kw = {"num_steps": 10, "v_reset": None, "v_th": 0.1, "neuron_type": kind, "surrogate_fun": "surrogate.ATan()",
          "tau": 2.0, "detach_reset": True, "spike_norm": "BN"}
    m = Spiking_BN_WindowAttention3D(96, (2, 9, 9), (0, 0, 0), 3, version="swinv1", norm="BN", **kw)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items() if not k.endswith("relative_position_index")})
    m.load_state_dict(sd, strict=False)
    m = m.eval().to(DEV)
    x = (rnd((2, 8, 9, 9, 96), 11) > 0.4).float().to(DEV)
    mask = compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4), DEV)
    y, _ = m(x, mask)
    ref = g[f"{kind}_y"]                                            # (B_, 162, C) spikes of the reference module
    assert (y.cpu().numpy().astype(np.uint8) != ref).mean() < 5e-4   # threshold-rounding flips only
    y2, _ = m(x, None)
    assert (y2.cpu().numpy().astype(np.uint8) != g[f"{kind}_y_nomask"]).mean() < 5e-4


def test_batch2_three_encoder_model_teacher_forced():
    """B = 2 (the reference couples batch elements through its raw reshapes) on the 3-encoder
    MS_SpikingformerFlowNet at 144x192: every stage must reproduce the oracle *at that batch size*."""
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet
    model, sd, ocfg = build("lif", 144, 192, MS_SpikingformerFlowNet)
    chunk = O.prepare_chunk(synth_voxel(2, 10, 144, 192, seed=77))
    teacher_forced_all("B=2", model.to(DEV).engine(), sd, ocfg, chunk)
    with torch.no_grad():
        # and the two samples really are coupled: sample 0 of the batch differs from sample 0 run alone
        alone = O.forward_flownet(chunk[:1], sd, ocfg)[-1]
        both = O.forward_flownet(chunk, sd, ocfg)[-1][:1]
        assert (alone - both).abs().max() > 1e-3
    out = model(chunk.to(DEV))
    assert len(out["flow"]) == 3 and out["flow"][-1].shape == (2, 2, 144, 192)


def test_long_T20_stage_parity():
    """BASELINE configs[4] flavour: 20 bins / T = 20 (long LIF scan, T = 20 GEMM epilogue, unfused conv path)."""
    model, sd, ocfg = build("lif", 144, 192, T=20)
    chunk = O.prepare_chunk(synth_voxel(1, 20, 144, 192, seed=78))
    teacher_forced_all("T=20", model.to(DEV).engine(), sd, ocfg, chunk, stages=1)


def test_mdr_config_window8_T5_psn():
    """The second shipped SNN config (configs/train_MDR_supervised_SDformerFlow.yml): window (2,8,8), num_steps 5,
    psn, 256x256 crops - no window padding, T = 5 epilogues, 64-token slices."""
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(CFG), "train_MDR_supervised_SDformerFlow.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"])
    cfg["swin_transformer"]["input_size"] = [256, 256]
    model = MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy())
    sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd, strict=True)
    model.eval()
    n = O.NeuronCfg("psn", cfg["spiking_neuron"]["v_th"], None, 2.0, 5)
    ocfg = {"neuron": n, "num_bins": 10, "window_size": (2, 8, 8), "depths": [2, 2, 6, 2], "num_heads": [3, 6, 12, 24]}
    chunk = O.prepare_chunk(synth_voxel(1, 10, 256, 256, seed=79))
    teacher_forced_all("MDR", model.to(DEV).engine(), sd, ocfg, chunk)
    with torch.no_grad():
        refs = O.forward_flownet(chunk, sd, ocfg)
    out = model(chunk.to(DEV))["flow"]
    assert len(out) == 4 and all(torch.isfinite(f).all() for f in out)
    assert abs(out[-1].abs().mean().item() - refs[-1].abs().mean().item()) < 0.2 * refs[-1].abs().mean().item()


def test_forwards_in_flight_on_streams_and_graph_replay_are_bit_equal():
    """bench.py keeps several independent forwards in flight (own HIP stream, own HIP graph): every one of them must be
    the plain forward bit for bit - separate split-K workspaces, no shared scratch, deterministic kernels."""
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet
    model, sd, ocfg = build("lif", 144, 192, MS_SpikingformerFlowNet)
    model = model.to(DEV)
    xs = [O.prepare_chunk(synth_voxel(1, 10, 144, 192, seed=200 + i)).to(DEV) for i in range(3)]
    with torch.no_grad():
        refs = [[f.clone() for f in model(x)["flow"]] for x in xs]
        streams = [torch.cuda.Stream() for _ in xs]
        graphs, outs = [], []
        for st, x in zip(streams, xs):
            with torch.cuda.stream(st):
                model(x)                                                 # this stream's workspace / plan caches
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                o = model(x)
            graphs.append(g); outs.append(o)
        for _ in range(3):                                               # all three in flight, several rounds
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
        torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert all(torch.equal(a, b) for a, b in zip(o["flow"], r))


def test_config5_full_size_T20_batch4_properties():
    """BASELINE configs[4] at FULL size: 20 bins / T = 20, 480 x 640, batch 4 (stage 0: 120 x 160 tokens padded to
    126 x 162, 10 080 windows; 80 images of 240 x 320 x 96 fp32 in the decoder, i.e. operands beyond the conv kernel's
    31-bit offsets, launched in image chunks).  The oracle needs minutes per sample at this size, so the checks are
    the size-independent ones: shapes, finiteness, run-to-run bit equality, batch coupling (sample 0
    inside a batch differs from sample 0 alone), and the chunked convolution against the same convolution launched
    on each half of the images separately (bit-equal)."""
    from sdformerflow_amd import hip
    model, sd, ocfg = build("lif", 480, 640, T=20)
    model = model.to(DEV)
    chunk = O.prepare_chunk(synth_voxel(4, 20, 480, 640, seed=1234 + 5)).to(DEV)
    with torch.no_grad():
        a = [f.clone() for f in model(chunk)["flow"]]
        b = model(chunk)["flow"]
        one = model(chunk[:1])["flow"]
    assert len(a) == 4 and all(f.shape == (4, 2, 480, 640) and torch.isfinite(f).all() for f in a)
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    assert not torch.equal(a[-1][:1], one[-1])
    assert 0.5 < a[-1].abs().mean().item() / one[-1].abs().mean().item() < 2.0
    # a convolution whose fp32 output (80 x 240 x 320 x 96 x 4 B = 2.36 GB) exceeds 2^31 bytes: chunked == per-half
    g = torch.Generator(device=DEV).manual_seed(3)
    imgs, H, W, Cin, Cout = 80, 240, 320, 48, 96
    x = (torch.rand((imgs, H, W, Cin), device=DEV, generator=g) < 0.2).to(torch.uint8)
    Wp = hip.pack_conv_weight(torch.randn((Cout, Cin, 3, 3), device=DEV, generator=g) * 0.05, 2)
    al, be = torch.rand(Cout, device=DEV, generator=g) + 0.5, torch.randn(Cout, device=DEV, generator=g) * 0.1
    full = torch.empty((imgs * H * W, Cout), device=DEV)
    hip.spike_conv2d(x, Wp, imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=full, alpha=al, beta=be)
    half = torch.empty((imgs // 2 * H * W, Cout), device=DEV)
    for h in range(2):
        hip.spike_conv2d(x[h * imgs // 2:(h + 1) * imgs // 2], Wp, imgs // 2, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1),
                         out=half, alpha=al, beta=be)
        assert torch.equal(full[h * imgs // 2 * H * W:(h + 1) * imgs // 2 * H * W], half)


def test_large_window_15x15_T20():
    """The large-window variant of config 5 (window (2,15,15), N = 450 tokens per window, the value left commented in the
    reference's configs/valid_DSEC_supervised.yml:18): stage 0 teacher-forced against the oracle at 120 x 160 input
    (one stage-0 window row), then the full 480 x 640 forward for shapes / finiteness / determinism."""
    cfg = yaml.safe_load(open(CFG))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type="lif", num_steps=20)
    cfg["model"]["num_bins"] = 20
    cfg["swin_transformer"].update(input_size=[120, 160], window_size=[2, 15, 15])
    model = MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy())
    sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd, strict=True)
    model = model.eval().to(DEV)
    n = O.NeuronCfg("lif", 0.1, None, 2.0, 20)
    chunk = O.prepare_chunk(synth_voxel(1, 20, 120, 160, seed=81))
    p = "sttmultires_unet.encoders.swin3d."
    eng = model.engine()
    ocfg = {"neuron": n, "num_bins": 20, "window_size": (2, 15, 15), "depths": [2, 2, 6, 2], "num_heads": [3, 6, 12, 24]}
    teacher_forced_all("win15", eng, sd, ocfg, chunk, stages=1)
    cfg["swin_transformer"].update(input_size=[480, 640])
    model2 = MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy())
    model2.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model2.state_dict().items()}), strict=True)
    model2 = model2.eval().to(DEV)
    big = O.prepare_chunk(synth_voxel(1, 20, 480, 640, seed=82)).to(DEV)
    with torch.no_grad():
        a = [f.clone() for f in model2(big)["flow"]]
        b = model2(big)["flow"]
    assert all(f.shape == (1, 2, 480, 640) and torch.isfinite(f).all() for f in a)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_unet_tail_with_mismatched_feature_sizes_takes_the_concat_path():
    """Odd feature maps (19x25 -> 10x13 -> 5x7): the up-sampled stream (10x14) is larger than the skip (10x13), the reference's
    skip_concat crops / pads it (models/model_util.py:14-19) - the engine's concatenation path (reference channel order);
    equal sizes take the concatenation-free path, which the teacher-forced stage test covers.  Teacher-forced on synthetic
    encoder features against the oracle's U-Net tail."""
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet
    from sdformerflow_amd.synthetic import synth_uniform
    model, sd, ocfg = build("lif", 144, 192, cls=MS_SpikingformerFlowNet)
    n = ocfg["neuron"]
    eng = model.to(DEV).engine()
    feats = [synth_uniform((1, 10, h, w, c), 50 + i, -0.4, 0.9) for i, (h, w, c) in enumerate(((19, 25, 96), (10, 13, 192), (5, 7, 384)))]
    report = []
    O_feats = [f.permute(1, 0, 4, 2, 3).contiguous() for f in feats]
    stage_replay("unet_tail(odd)", eng, lambda: [pp.permute(1, 0, 4, 2, 3) for pp in eng.unet_tail([f.contiguous().to(DEV) for f in feats])],
                 lambda: unet_tail_oracle(O_feats, sd, n), report)
    print("odd sizes", report[0][1], "output max-dev %.1e" % report[0][2])


def test_feature_map_smaller_than_the_window_is_refused():
    """At 96 x 128 the third stage of the en4 model is 6 x 8 tokens, smaller than the (2, 9, 9) window: the reference clamps the
    window (Spiking_swin_transformer3D.py:786) and then fails to view its positional encoding as (T', 1, Wh, Ww, C) (:678); the
    engine refuses the input as well instead of reading the table with the wrong pitch."""
    from sdformerflow_amd import hip
    from sdformerflow_amd.harness import prepare_chunk
    model, _, _ = build("lif", 96, 128)
    x = prepare_chunk(synth_voxel(1, 10, 96, 128, seed=3)).to("cuda:0")
    with pytest.raises(hip.SdfError, match="smaller than the window"):
        model.to("cuda:0")(x)
