"""ANN STTFlowNet (BASELINE config 3): the model-level caller of the fused window-attention kernel (section 8 row a10).
CPU: the module tree has the reference's state_dict schema.  GPU: flows vs the CPU oracle (itself pinned to the
reference by tests/test_oracle_golden.py::test_ann_sttflownet_end_to_end_matches_reference) and vs the golden file."""
import os

import numpy as np
import pytest
import torch
import yaml

from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def build(cls_name="STTFlowNet", size=(144, 192)):
    from sdformerflow_amd.STSwinNet import STSwinNet
    cfg = yaml.safe_load(open(os.path.join(ROOT, "sdformerflow_amd", "configs", "train_DSEC_supervised_STT_voxel.yml")))
    model = dict(cfg["model"], spiking_neuron=None)
    swin = dict(cfg["swin_transformer"], input_size=list(size))
    if cls_name == "STTFlowNet_4en":
        swin.update(swin_depths=[2, 2, 6, 2], swin_num_heads=[3, 6, 12, 24], swin_out_indices=[0, 1, 2, 3])
    return getattr(STSwinNet, cls_name)(model, swin)


def load_synth(net):
    skip = ("relative_position_index", "relative_coords_table", "num_batches_tracked")
    sd = synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith(skip)})
    net.load_state_dict(sd, strict=False)
    return sd


def test_sttflownet_state_schema_matches_reference():
    net = build()
    mine = [(k, "x".join(str(d) for d in v.shape)) for k, v in net.state_dict().items()]
    ref = []
    with open(os.path.join(G, "state_schema_sttflownet.txt")) as f:
        for line in f:
            n, _, shp = line.strip().partition(" ")
            ref.append((n, shp))
    assert mine == ref


def test_sttflownet_refuses_cpu_and_training():
    from sdformerflow_amd import hip
    net = build().eval()
    with pytest.raises(hip.SdfError):
        net(torch.zeros(1, 20, 144, 192), None)
    net.train()
    with pytest.raises(NotImplementedError):
        net(torch.zeros(1, 20, 144, 192), None)


@pytest.mark.gpu
def test_sttflownet_flows_match_oracle_and_golden():
    from oracle import sdformer_oracle as O
    net = build().eval()
    sd = load_synth(net)
    vox = synth_voxel(2, 20, 144, 192, seed=1237)
    cfg = {"num_bins": 20, "patch_size": (10, 4, 4), "window_size": (2, 9, 9), "depths": [2, 2, 6], "num_heads": [3, 6, 12]}
    with torch.no_grad():
        ref = O.forward_sttflownet(vox, sd, cfg)
    net = net.cuda()
    net.norm_input = False                                       # the fixture was generated on the single-chunk input
    got = net(vox.cuda(), None)["flow"]
    g = np.load(os.path.join(G, "ann_end_to_end.npz"))
    assert len(got) == len(ref) == 3
    for i, (a, b) in enumerate(zip(got, ref)):
        a = a.cpu()
        tol = 1e-3 * b.abs().mean().item()                       # north star: flow within 1e-3 relative (fp32)
        assert (a - b).abs().max().item() <= tol, (i, (a - b).abs().max().item(), tol)
        s = a.shape[-1] // (24 * 2 ** i)
        assert np.abs(a[:, :, ::s, ::s].numpy() - g[f"flow{i}"]).max() <= tol


@pytest.mark.gpu
def test_sttflownet_4en_matches_oracle():
    from oracle import sdformer_oracle as O
    net = build("STTFlowNet_4en", (288, 384)).eval()
    sd = load_synth(net)
    vox = synth_voxel(1, 20, 288, 384, seed=77)           # BASELINE config 3 crop; 4 scales need /32
    cfg = {"num_bins": 20, "patch_size": (10, 4, 4), "window_size": (2, 9, 9), "depths": [2, 2, 6, 2], "num_heads": [3, 6, 12, 24]}
    with torch.no_grad():
        ref = O.forward_sttflownet(vox, sd, cfg)
    net = net.cuda()
    net.norm_input = False
    got = net(vox.cuda(), None)["flow"]
    assert len(got) == len(ref) == 4
    for i, (a, b) in enumerate(zip(got, ref)):
        d = (a.cpu() - b).abs().max().item()
        assert d <= 1e-3 * b.abs().mean().item(), (i, d)


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [(0, 0, 0), (1, 4, 4)])
def test_window_partition_inside_the_attention_kernel_equals_the_materialised_sequence(shift):
    """`SwinTransformerBlock3D.forward` reads / writes the attention's rows through the slice map inside the kernel; the
    reference's pad + roll + window_partition ... window_reverse + roll + crop sequence (swin_transformer3D_v2.py:286-310) is
    kept as the `SDF_ATTN_MATERIALISE=1` A/B path.  Odd sizes (padding tokens read the qkv bias and write nothing), shift with
    mask, batch 2; equal up to the library GEMM's row-count-dependent rounding (the projections run on different row orders)."""
    from sdformerflow_amd.STSwinNet.swin_transformer3D_v2 import SwinTransformerBlock3D
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    blk = SwinTransformerBlock3D(96, 3, window_size=(2, 9, 9), shift_size=shift, qkv_bias=True)
    sd = synth_state_dict({k: tuple(v.shape) for k, v in blk.state_dict().items()
                           if not k.endswith(("relative_position_index", "relative_coords_table"))})
    blk.load_state_dict(sd, strict=False)
    blk = blk.to("cuda:0").eval()
    x = rnd((2, 3, 20, 25, 96), 61, -1.0, 1.0).to("cuda:0")              # D = 3 -> 4, H = 20 -> 27, W = 25 -> 27 padded
    with torch.no_grad():
        fused = blk(x)
        with hip.scoped_switches(SDF_ATTN_MATERIALISE="1"):
            ref = blk(x)
    assert fused.shape == ref.shape == x.shape
    assert (fused - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.gpu
def test_config3_batch8_dense_convolution_path(monkeypatch):
    """BASELINE configs[2] at full size (batch 8, 20 bins, 288 x 384): the patch embedding's stride-1 convolutions on this
    framework's dense convolution (two fp16 planes per operand, csrc/dense_conv_wres.hip) against (a) the CPU oracle on the first
    and last sample and (b) the same network with the library's fp32 convolutions and GEMMs on the whole batch - the 1e-3 bound of
    the north star, and in practice well over an order closer to the library path than that.  The swin blocks' Linear layers run on
    csrc/dense_linear.hip in the first pass as well."""
    from oracle import sdformer_oracle as O
    net = build("STTFlowNet", (288, 384)).eval()
    sd = load_synth(net)
    vox = synth_voxel(8, 20, 288, 384, seed=1237)
    cfg = {"num_bins": 20, "patch_size": (10, 4, 4), "window_size": (2, 9, 9), "depths": [2, 2, 6], "num_heads": [3, 6, 12]}
    net = net.cuda()
    net.norm_input = False
    calls = []
    from sdformerflow_amd import hip
    real = hip.dense_conv3x3
    monkeypatch.setattr(hip, "dense_conv3x3", lambda *a, **k: (calls.append(a[0].shape), real(*a, **k))[1])
    got = [f.cpu() for f in net(vox.cuda(), None)["flow"]]
    # head + 8 residual-block convolutions on 16 images (batch x 2 temporal chunks); the three decoders' 768 / 386 / 194-channel
    # convolutions as chains of 8 / 5 / 3 slices on 8 images
    assert [c[0] for c in calls] == [16] * 9 + [8] * 16 and [c[1] for c in calls[9:]] == [6] * 8 + [6] * 4 + [1] + [6] * 2 + [1], calls
    lin_calls = []
    real_lin = hip.dense_linear
    monkeypatch.setattr(hip, "dense_linear", lambda *a, **k: (lin_calls.append(a[0].shape), real_lin(*a, **k))[1])
    monkeypatch.setenv("SDF_DENSE_CONV", "0")                     # library convolutions and library GEMMs for the whole network
    monkeypatch.setenv("SDF_DENSE_LINEAR", "0")
    lib = [f.cpu() for f in net(vox.cuda(), None)["flow"]]
    assert len(calls) == 25 and not lin_calls
    monkeypatch.delenv("SDF_DENSE_LINEAR")
    net(vox.cuda(), None)
    # qkv, proj, fc1, fc2 of the 10 blocks + the 2 patch-merging reductions, less all four of the first stage's two blocks: their
    # attention half and their MLP half are one launch each with both products inside (csrc/ann_block.hip, csrc/ann_mlp_block.hip)
    assert len(lin_calls) == 4 * (2 + 2 + 6) + 2 - 2 * 4, len(lin_calls)
    for i, (a, b) in enumerate(zip(got, lib)):
        d = (a - b).abs().max().item()
        assert d <= 5e-5 * b.abs().mean().item(), (i, d, b.abs().mean().item())
    for n in (0, 7):
        with torch.no_grad():
            ref = O.forward_sttflownet(vox[n:n + 1], sd, cfg)
        for i, (a, b) in enumerate(zip(got, ref)):
            d = (a[n:n + 1] - b).abs().max().item()
            assert d <= 1e-3 * b.abs().mean().item(), (n, i, d)


@pytest.mark.gpu
def test_patch_embedding_and_decoder_modules_match_their_library_forms(monkeypatch):
    """Module level: `PatchEmbedLocal` (dense convolutions + gathered-row projection) and `UpsampleConvLayer.forward_parts` (fused
    upsample-packing + slice chain) against the same modules on the library's fp32 convolutions, fed the same tensors."""
    from sdformerflow_amd.STSwinNet.PatchEmbed import PatchEmbedLocal
    from sdformerflow_amd.STSwinNet.STSwinNet import UpsampleConvLayer
    g = torch.Generator().manual_seed(21)
    pe = PatchEmbedLocal((48, 64), (10, 4, 4), 20, 96).eval()
    with torch.no_grad():
        for p in pe.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.2) + (1.0 if p.dim() == 1 and p.shape[0] == 96 else 0.0))
        for n, b in pe.named_buffers():
            if n.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
            elif n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.1)
    pe = pe.cuda()
    x = torch.randn(2, 3, 10, 48, 64, generator=g).cuda()
    dec = UpsampleConvLayer(2 * 96 + 2, 96, 3).eval().cuda()
    parts = [torch.randn(3, 96, 12, 16, generator=g).cuda(), torch.randn(3, 96, 12, 16, generator=g).cuda(), torch.randn(3, 2, 12, 16, generator=g).cuda()]
    with torch.no_grad():
        y = pe(x)
        z = dec.forward_parts(parts, [2, 0, 1])
        monkeypatch.setenv("SDF_DENSE_CONV", "0")
        y_lib = pe(x)
        z_lib = dec.forward_parts(parts, [2, 0, 1])
    assert y.shape == y_lib.shape == (3, 96, 2, 12, 16) and z.shape == z_lib.shape == (3, 96, 24, 32)
    assert (y - y_lib).abs().max().item() <= 2e-5 * y_lib.abs().max().item()
    assert (z - z_lib).abs().max().item() <= 2e-5 * z_lib.abs().max().item()


@pytest.mark.gpu
def test_feature_map_smaller_than_the_window_is_refused_like_the_reference():
    """At 96 x 128 the third stage is 6 x 8 tokens, smaller than the (2, 9, 9) window: the reference clamps the window and then
    raises when it adds the 162 x 162 position bias (swin_transformer3D_v2.py:190); this build raises too (it used to run the
    attention kernel past its row map)."""
    net = build("STTFlowNet", (96, 128)).eval()
    load_synth(net)
    net = net.cuda()
    with pytest.raises(RuntimeError, match="smaller than the window"):
        net(synth_voxel(1, 20, 96, 128, seed=5).cuda(), None)


@pytest.mark.gpu
def test_odd_feature_sizes_follow_skip_concat():
    """150 x 200: the stage sizes are 38 x 50, 19 x 25, 10 x 13, so every decoder output is one larger than its skip and
    `skip_concat` (models/model_util.py:14-19, used at models/STSwinNet/STSwinNet.py:277-279) crops it before the concatenation."""
    from oracle import sdformer_oracle as O
    net = build("STTFlowNet", (150, 200)).eval()
    sd = load_synth(net)
    vox = synth_voxel(1, 20, 150, 200, seed=9)
    cfg = {"num_bins": 20, "patch_size": (10, 4, 4), "window_size": (2, 9, 9), "depths": [2, 2, 6], "num_heads": [3, 6, 12]}
    with torch.no_grad():
        ref = O.forward_sttflownet(vox, sd, cfg)
    net = net.cuda()
    net.norm_input = False
    got = net(vox.cuda(), None)["flow"]
    for i, (a, b) in enumerate(zip(got, ref)):
        assert a.shape == b.shape
        d = (a.cpu() - b).abs().max().item()
        assert d <= 1e-3 * b.abs().mean().item(), (i, d)
