"""Pin the CPU oracle (oracle/sdformer_oracle.py) against the golden fixtures that
tests/golden/make_golden.py produced by running the real reference.  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import sdformer_oracle as O
from sdformerflow_amd.synthetic import synth_state_dict, synth_uniform as rnd, synth_voxel, synth_label

G = os.path.join(os.path.dirname(__file__), "golden")


def gold(name):
    return np.load(os.path.join(G, name + ".npz"))


def ncfg(kind, T):
    return O.NeuronCfg(kind, v_th=0.1, v_reset=None, tau=2.0, num_steps=T)


def qk_shapes(C, nH, kind, T=2):
    s = {"positional_encoding": (1, nH, 162, C // nH), "linear_q.weight": (C, C), "linear_k.weight": (C, C),
         "proj.weight": (C, C), "proj.bias": (C,)}
    for bn in ("bn_q", "bn_k", "proj_bn"):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{bn}.norm_layer.{leaf}"] = (C,)
    if kind == "psn":
        for sn in ("sn_q", "sn_k", "sn2_q", "attn_sn", "proj_sn"):
            s[f"{sn}.spiking_neuron.weight"] = (T, T)
            s[f"{sn}.spiking_neuron.bias"] = (T, 1)
    return s


def neuron_input(T):
    x = rnd((T, 4096), 100 + T, -0.3, 0.6)
    x[:, :64] = 0.1
    x[0, 64:128] = 0.2
    return x


# ---------------------------------------------------------------- neurons (a1, a2)
@pytest.mark.parametrize("T", [2, 10, 20])
def test_lif_matches_reference_stub(T):
    g = gold("neurons")
    x = neuron_input(T)
    for tag, vr in (("soft", None), ("hard", 0.0)):
        s, v = O.lif_multistep(x, 2.0, 0.1, vr, return_v=True)
        assert np.array_equal(s.numpy().astype(np.uint8), g[f"lif_{tag}_T{T}_s"])      # bit-exact spikes
        assert np.array_equal(v.numpy(), g[f"lif_{tag}_T{T}_v"])                        # bit-exact membrane


@pytest.mark.parametrize("T", [4, 10])
def test_plif_slttlif_glif_match_the_reference_switch(T):
    """The neuron types no shipped configuration uses (reference Spiking_modules.py:49-56, 75-92; fixture = the reference's own
    `Spiking_neuron` in multi-step mode): python oracle and C oracle for plif / SLTTlif, the oracle's restatement of the GLIF recurrence (the product's module runs on the GPU
    only: tests/test_hip_kernels.py)."""
    from oracle import neuron_ref as R
    from sdformerflow_amd.STSwinNet_SNN.Spiking_modules import Spiking_neuron
    g = gold("neurons_extra")
    x = torch.from_numpy(g[f"x_T{T}"])
    k = float(torch.sigmoid(torch.from_numpy(g[f"plif_T{T}_w"])))
    assert 0.0 < k < 1.0 and k != 0.5
    for tag, vr in (("soft", None), ("hard", 0.0), ("hard05", 0.05)):
        for kind, tau in (("plif", k), ("SLTTlif", 2.0)):
            if f"{kind}_{tag}_T{T}_s" not in g:
                continue
            want = g[f"{kind}_{tag}_T{T}_s"]
            assert np.array_equal(O.lif_multistep(x, tau, 0.1, vr).numpy().astype(np.uint8), want), (kind, tag)
            assert np.array_equal(R.neuron_ref(x, "lif", tau, 0.1, vr).numpy().astype(np.uint8), want), (kind, tag)
            sd = {"n.w": torch.from_numpy(g[f"plif_T{T}_w"])}
            assert np.array_equal(O.neuron(x, O.NeuronCfg(kind, 0.1, vr, 2.0, T), sd, "n.").numpy().astype(np.uint8), want)
    gsd = {kk[len(f"glif_T{T}/"):]: torch.from_numpy(v) for kk, v in g.items() if kk.startswith(f"glif_T{T}/")}
    got = O.glif_multistep(3.0 * x, gsd, "spiking_neuron.")
    assert np.array_equal(got.numpy().astype(np.uint8), g[f"glif_T{T}_s"]) and 0.05 < got.mean() < 0.95
    m = Spiking_neuron(num_steps=T, neuron_type="glif").eval()                   # the product module: same schema, GPU tensors only
    m.load_state_dict(gsd)
    with pytest.raises(Exception, match="GPU"):
        m(x)
    with pytest.raises(NotImplementedError):
        m.train()(x)                                                             # inference only, loudly


@pytest.mark.parametrize("T", [2, 10, 20])
def test_psn_matches_reference(T):
    g = gold("neurons")
    x = neuron_input(T)
    w, b = torch.from_numpy(g[f"psn_T{T}_w"]), torch.from_numpy(g[f"psn_T{T}_b"])
    h = O.psn_h(x, w, b)
    href = g[f"psn_T{T}_h"]
    # torch.addmm's fp32 order is BLAS-defined; the oracle's fixed-order fmaf-chain H must agree to fp32 rounding
    assert np.abs(h.numpy() - href).max() <= 4e-7 * max(1.0, np.abs(href).max())
    s = O.psn(x, w, b).numpy().astype(np.uint8)
    diff = s != g[f"psn_T{T}_s"]
    # spikes may only differ where the reference's own H is within rounding of the threshold
    assert np.all(np.abs(href[diff]) < 1e-6)
    assert diff.mean() < 1e-4


# ---------------------------------------------------------------- window geometry (a4, a6, a11)
@pytest.mark.parametrize("tag", ["s0", "odd", "noshift"])
def test_slice_table_matches_reference(tag):
    g = gold("index_maps")
    B, D, H, W, w0, w1, w2, s0, s1, s2 = [int(v) for v in g[f"{tag}_shape"]]
    ws, ss = O.get_window_size((D, H, W), (w0, w1, w2), (s0, s1, s2))
    src, B_ = O.slice_table(B, D, H, W, ws, ss)
    assert bool(g[f"{tag}_roundtrip_ok"])
    assert np.array_equal(src.astype(np.int32), g[f"{tag}_gather"])
    # scatter is the exact inverse on valid positions
    x = rnd((B, D, H, W, 3), 5)
    y, _ = O.gather_slices(x, ws, ss)
    assert torch.equal(O.scatter_slices(y, B, D, H, W, ws, ss), x)
    Dp, Hp, Wp = -(-D // ws[0]) * ws[0], -(-H // ws[1]) * ws[1], -(-W // ws[2]) * ws[2]
    m = O.compute_mask(Dp, Hp, Wp, ws, ss)
    assert float(m.sum()) == float(g[f"{tag}_mask_sum"])
    if tag != "s0":
        assert np.array_equal(m.numpy().astype(np.int8), g[f"{tag}_mask"])


# ---------------------------------------------------------------- a5
@pytest.mark.parametrize("tag,kind", [("c96_lif", "lif"), ("c96_psn", "psn"), ("c192_lif", "lif"), ("c384_psn", "psn")])
def test_qk_attention_matches_reference(tag, kind):
    g = gold("qk_attention")
    B_, C, nH, seed = [int(v) for v in g[f"{tag}_cfg"]]
    sd = synth_state_dict(qk_shapes(C, nH, kind))
    x = rnd((2, B_, 9, 9, C), seed, -0.5, 1.0).reshape(2, B_, 81, C)
    y, e, z = O.qk_attention(x, sd, "", nH, ncfg(kind, 2))
    yref = g[f"{tag}_y"]                                                  # (B_,162,C) == raw reshape
    assert np.abs(y.reshape(B_, 162, C).numpy() - yref).max() <= 2e-5
    # explicit gather table == the reshape/permute formulation
    tab = torch.from_numpy(O.z_gather_table(B_, nH, 2, 81, C // nH))
    assert torch.equal(e.reshape(-1)[tab.reshape(-1)].view_as(z), z)
    # the module's second return value (`attn_sn` on the gated tensor: what log=True collects), bit for bit
    gs = gold("qk_attention_scores")
    want = np.unpackbits(gs[f"{tag}_attn"])[:2 * B_ * 81 * C].reshape(2, B_, 81, C)
    score = O.attention_score(z, sd, "", ncfg(kind, 2))
    assert tuple(gs[f"{tag}_shape"]) == (2, B_, 9, 9, C) and np.array_equal(score.numpy().astype(np.uint8), want)
    assert 0.03 < score.mean() < 0.97


# ---------------------------------------------------------------- a9
def sew_shapes(C, nH, kind, T=2):
    s = {"relative_position_bias_table": (3 * 17 * 17, nH), "proj.weight": (C, C), "proj.bias": (C,)}
    for n in ("q", "k", "v"):
        s[f"linear_{n}.weight"] = (C, C)
    for bn in ("bn_q", "bn_k", "bn_v", "proj_bn"):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{bn}.norm_layer.{leaf}"] = (C,)
    if kind == "psn":
        for sn in ("sn_q", "sn_k", "sn_v", "attn_sn", "proj_sn"):
            s[f"{sn}.spiking_neuron.weight"] = (T, T)
            s[f"{sn}.spiking_neuron.bias"] = (T, 1)
    return s


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_sew_attention_matches_reference(kind):
    g = gold("sew_attention")
    C, nH, B, nW = 96, 3, 2, 4
    sd = synth_state_dict(sew_shapes(C, nH, kind))
    x = (rnd((2, B * nW, 9, 9, C), 11) > 0.4).float().reshape(2, B * nW, 81, C)
    mask = O.compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4))
    y, attn = O.sew_attention(x, sd, "", nH, (2, 9, 9), ncfg(kind, 2), mask)
    assert np.abs(attn[0, 0].numpy() - g[f"{kind}_attn00"]).max() <= 1e-4
    assert np.abs(attn.sum((-1, -2)).numpy() - g[f"{kind}_attn_sum"]).max() <= 0.5
    yref = g[f"{kind}_y"].reshape(2, B * nW, 81, C)
    assert (y.numpy().astype(np.uint8) != yref).mean() < 2e-4           # threshold-rounding flips only
    y2, _ = O.sew_attention(x, sd, "", nH, (2, 9, 9), ncfg(kind, 2), None)
    assert (y2.numpy().astype(np.uint8) != g[f"{kind}_y_nomask"].reshape(2, B * nW, 81, C)).mean() < 2e-4


# ---------------------------------------------------------------- a10
def test_ann_attention_matches_reference():
    g = gold("ann_attention")
    C, nH, B, nW = 96, 3, 1, 4
    shapes = {"logit_scale": (nH, 1, 1), "cpb_mlp.0.weight": (512, 3), "cpb_mlp.0.bias": (512,),
              "cpb_mlp.2.weight": (nH, 512), "qkv.weight": (3 * C, C), "qkv.bias": (3 * C,),
              "proj.weight": (C, C), "proj.bias": (C,)}
    sd = synth_state_dict(shapes)
    assert np.allclose(O.relative_coords_table((2, 9, 9)).numpy(), g["coords_table"], atol=1e-6)
    assert int(O.relative_position_index((2, 9, 9)).sum()) == int(g["rel_index_sum"])
    x = rnd((B * nW, 162, C), 13, -1.0, 1.0)
    mask = O.compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4))
    y, attn = O.ann_window_attention(x, sd, "", nH, (2, 9, 9), mask)
    assert np.abs(attn[0, 0].numpy() - g["attn00"]).max() <= 1e-6
    assert np.abs(y.numpy() - g["y"]).max() <= 1e-5
    y2, _ = O.ann_window_attention(x, sd, "", nH, (2, 9, 9), None)
    assert np.abs(y2.numpy() - g["y_nomask"]).max() <= 1e-5


# ---------------------------------------------------------------- a7, a8
def block_shapes(C, nH, kind, T):
    s = {"attn." + k: v for k, v in qk_shapes(C, nH, kind).items()}
    s.update({"mlp.fc1.weight": (4 * C, C), "mlp.fc2.weight": (C, 4 * C)})
    for bn, n in (("mlp.bn1", 4 * C), ("mlp.bn2", C)):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"{bn}.norm_layer.{leaf}"] = (n,)
    if kind == "psn":
        for sn in ("mlp.sn1", "mlp.sn2"):
            s[f"{sn}.spiking_neuron.weight"] = (T, T)
            s[f"{sn}.spiking_neuron.bias"] = (T, 1)
    return s


@pytest.mark.parametrize("tag,kind", [("lif_sw", "lif"), ("lif_w", "lif"), ("psn_sw", "psn")])
def test_ms_block_matches_reference(tag, kind):
    g = gold("ms_block")
    H, W, s0, s1, s2 = [int(v) for v in g[f"{tag}_cfg"]]
    C, nH, T = 96, 3, 4
    sd = synth_state_dict(block_shapes(C, nH, kind, T))
    x = rnd((1, T, H, W, C), 17, -0.5, 1.0)
    y = O.ms_block(x, sd, "", nH, (2, 9, 9), (s0, s1, s2), ncfg(kind, T))
    d = np.abs(y.numpy() - g[f"{tag}_y"])
    assert d.max() <= 5e-5, d.max()


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_patch_merge_matches_reference(kind):
    g = gold("ms_block")
    C, T = 96, 4
    s = {"reduction.weight": (2 * C, 4 * C)}
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        s[f"norm.norm_layer.{leaf}"] = (2 * C,)
    if kind == "psn":
        s["sn.spiking_neuron.weight"], s["sn.spiking_neuron.bias"] = (T, T), (T, 1)
    sd = synth_state_dict(s)
    y = O.ms_patch_merge(rnd((1, T, 9, 21, C), 19, -0.5, 1.0), sd, "", ncfg(kind, T))
    assert np.abs(y.numpy() - g[f"{kind}_merge_y"]).max() <= 2e-5


# ---------------------------------------------------------------- end to end (BASELINE config 1)
def load_schema(kind):
    shapes = {}
    with open(os.path.join(G, f"state_schema_en4_{kind}.txt")) as f:
        for line in f:
            name, _, shp = line.strip().partition(" ")
            shapes[name] = tuple(int(v) for v in shp.split("x")) if shp else ()
    return shapes


def en4_cfg(kind):
    return {"neuron": ncfg(kind, 10), "num_bins": 10, "window_size": (2, 9, 9), "depths": [2, 2, 6, 2],
            "num_heads": [3, 6, 12, 24]}


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_end_to_end_flow_matches_reference(kind):
    g = gold("end_to_end")
    shapes = load_schema(kind)
    assert len(shapes) == int(g[f"{kind}_n_state"])
    sd = synth_state_dict({k: v for k, v in shapes.items() if not k.endswith("num_batches_tracked")})
    chunk = O.prepare_chunk(synth_voxel(1, 10, 288, 384, seed=1235))
    assert abs(float(chunk.double().sum()) - float(g[f"{kind}_chunk_checksum"])) < 1e-6
    O.PSN_MODE = "addmm"          # the reference's literal op: the net is chaotic, 1 flipped spike decorrelates it
    O.RATE_LOG = []
    try:
        with torch.no_grad():
            flows = O.forward_flownet(chunk, sd, en4_cfg(kind))
        rates = dict(O.RATE_LOG)
    finally:
        O.PSN_MODE, O.RATE_LOG = "fmaf", None
    # (1) robust on any host: per-layer firing rates (the dead `attn_sn` call is not on the forward path)
    ref_rates = {str(n) + ".": float(r) for n, r in zip(g[f"{kind}_rate_names"], g[f"{kind}_rates"])
                 if "attn_sn" not in str(n)}
    assert set(ref_rates) == set(rates) and len(rates) == 93
    assert max(abs(rates[k] - ref_rates[k]) for k in rates) < 2e-3
    # (2) bit-level: same ATen/MKL build as the generator => flows within the north-star 1e-3 bound.
    # On another CPU/BLAS the 1-ulp differences decorrelate the (chaotic) random-weight net, so the
    # strict check is only meaningful when nothing upstream moved.
    strict = all(abs(rates[k] - ref_rates[k]) < 1e-7 for k in rates)
    for i, f in enumerate(flows):
        s = f.shape[-1] // (24 * 2 ** i)
        ref = g[f"{kind}_flow{i}"]
        d = np.abs(f[:, :, ::s, ::s].numpy() - ref)
        scale = np.abs(ref).mean()
        if strict:
            assert d.max() <= 1e-3 * scale, (i, d.max(), scale)
        else:
            assert abs(np.abs(f.numpy()).mean() - float(g[f"{kind}_flow{i}_abs_mean"])) < 0.1 * scale
    label, mask = synth_label(1, 288, 384)
    m = O.aee(flows[-1], label, mask, 1.0)
    got = np.array([float(v.reshape(-1)[0]) for v in m])
    assert np.allclose(got, g[f"{kind}_aee"], rtol=1e-5 if strict else 5e-2, atol=1e-6 if strict else 5e-2)


# ---------------------------------------------------------------- ANN STTFlowNet end to end (BASELINE config 3 family)
def test_ann_sttflownet_end_to_end_matches_reference():
    g = gold("ann_end_to_end")
    shapes = {}
    with open(os.path.join(G, "state_schema_sttflownet.txt")) as f:
        for line in f:
            name, _, shp = line.strip().partition(" ")
            shapes[name] = tuple(int(v) for v in shp.split("x")) if shp else ()
    assert len(shapes) == int(g["n_state"])
    sd = synth_state_dict({k: v for k, v in shapes.items()
                           if not k.endswith(("relative_position_index", "relative_coords_table", "num_batches_tracked"))})
    cfg = {"num_bins": 20, "patch_size": (10, 4, 4), "window_size": (2, 9, 9), "depths": [2, 2, 6], "num_heads": [3, 6, 12]}
    with torch.no_grad():
        flows = O.forward_sttflownet(synth_voxel(2, 20, 144, 192, seed=1237), sd, cfg)
    for i, f in enumerate(flows):
        s = f.shape[-1] // (24 * 2 ** i)
        ref = g[f"flow{i}"]
        d = np.abs(f[:, :, ::s, ::s].numpy() - ref)
        assert d.max() <= 1e-3 * np.abs(ref).mean(), (i, d.max(), np.abs(ref).mean())      # north-star bound (ANN is not chaotic)


def test_ann_sttflownet_at_odd_feature_sizes_matches_reference():
    """150 x 200 (stage sizes 38 x 50, 19 x 25, 10 x 13): the decoders' `skip_concat` crops the upsampled features to the skip's
    size (models/model_util.py:14-19); fixture from the real reference (make_golden.py `ann_odd_size`)."""
    g = gold("ann_odd_size")
    shapes = {}
    with open(os.path.join(G, "state_schema_sttflownet.txt")) as f:
        for line in f:
            name, _, shp = line.strip().partition(" ")
            shapes[name] = tuple(int(v) for v in shp.split("x")) if shp else ()
    sd = synth_state_dict({k: v for k, v in shapes.items()
                           if not k.endswith(("relative_position_index", "relative_coords_table", "num_batches_tracked"))})
    cfg = {"num_bins": 20, "patch_size": (10, 4, 4), "window_size": (2, 9, 9), "depths": [2, 2, 6], "num_heads": [3, 6, 12]}
    with torch.no_grad():
        flows = O.forward_sttflownet(synth_voxel(1, 20, 150, 200, seed=9), sd, cfg)
    for i, f in enumerate(flows):
        ref = g[f"flow{i}"]
        assert tuple(f.shape) == (1, 2, 150, 200)
        d = np.abs(f[:, :, ::2, ::2].numpy() - ref)
        assert d.max() <= 1e-3 * np.abs(ref).mean(), (i, d.max(), np.abs(ref).mean())


# ------------------------------------------------------------------ neuron backward (training path)
NG = np.load(os.path.join(os.path.dirname(__file__), "golden", "neuron_grads.npz"))
LIF_GRAD_CASES = [("soft_detach", None, True, 2.0), ("soft_nodetach", None, False, 2.0), ("hard_detach", 0.0, True, 2.0),
                  ("hard_nodetach", 0.0, False, 2.0), ("tau3_soft_detach", None, True, 3.0)]


def neuron_grad_inputs(T):
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    x = rnd((T, 2048), 300 + T, -0.3, 0.6)
    x[:, :64] = 0.1
    x[0, 64:128] = 0.2
    return x, rnd((T, 2048), 400 + T, -1.0, 2.0)


@pytest.mark.parametrize("T", [2, 10])
@pytest.mark.parametrize("tag,v_reset,detach,tau", LIF_GRAD_CASES)
def test_oracle_lif_backward_matches_reference_autograd(T, tag, v_reset, detach, tau):
    from oracle import neuron_bwd_ref as B
    x, g = neuron_grad_inputs(T)
    _, s = B.lif_forward_h(x, tau, 0.1, v_reset)
    assert np.array_equal(s.numpy().astype(np.uint8), NG[f"lif_{tag}_T{T}_s"])
    gx = B.lif_backward(x, g, tau, 0.1, v_reset, detach, 2.0)
    ref = torch.from_numpy(NG[f"lif_{tag}_T{T}_gx"])
    if detach and tau == 2.0:
        assert torch.equal(gx, ref)                               # two-term sums only: bit-equal to autograd
    else:
        assert (gx - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()


@pytest.mark.parametrize("T", [2, 10])
def test_oracle_psn_backward_matches_reference_autograd(T):
    from oracle import neuron_bwd_ref as B
    from sdformerflow_amd.synthetic import synth_state_dict
    x, g = neuron_grad_inputs(T)
    sd = synth_state_dict({"spiking_neuron.weight": (T, T), "spiking_neuron.bias": (T, 1)}, salt=T)
    gx, gW, gb = B.psn_backward(x, sd["spiking_neuron.weight"], sd["spiking_neuron.bias"], g, 2.0)
    for got, key in ((gx, "gx"), (gW, "gW"), (gb, "gb")):
        ref = torch.from_numpy(NG[f"psn_T{T}_{key}"])
        assert got.shape == ref.shape
        assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item(), key


# ------------------------------------------------------------------ train mode (BASELINE config 4)
TB = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_block.npz"))
TS = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_step.npz"))


def _train_sd(shapes_from, salt=0):
    """Seeded weights as leaf tensors that require grad (buffers stay plain)."""
    from sdformerflow_amd.synthetic import synth_state_dict
    sd = synth_state_dict(shapes_from, salt)
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))
                else v.clone()) for k, v in sd.items()}


def _close(got, ref, rate_tol, what):
    got, ref = got.detach().float(), torch.as_tensor(ref).float()
    scale = ref.abs().mean().item() + 1e-12
    bad = (got - ref).abs() > 1e-3 * scale
    assert bad.float().mean().item() <= rate_tol, (what, bad.float().mean().item())


@pytest.mark.parametrize("tag,kind", [("lif_sw", "lif"), ("psn_w", "psn")])
def test_oracle_train_mode_block_matches_reference(tag, kind):
    """TRAIN-mode MS swin block: output, dL/dx, every parameter gradient and the BN running-stat updates against the
    reference's own autograd (tests/golden/make_golden.py `train_block`)."""
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    B, H, W, *shift = (int(v) for v in TB[f"{tag}_cfg"])
    shapes = {k[len(tag) + 3:]: tuple(TB[k].shape) for k in TB.files if k.startswith(tag + "_g/")}
    shapes.update({k[len(tag) + 3:]: tuple(TB[k].shape) for k in TB.files if k.startswith(tag + "_r/")})
    for k in list(shapes):
        if k.endswith("running_mean"):
            shapes[k.replace("running_mean", "weight")] = shapes[k]
            shapes[k.replace("running_mean", "bias")] = shapes[k]
    if kind == "psn":
        for k in [k for k in shapes if k.endswith("spiking_neuron.weight")]:
            shapes[k.replace("weight", "bias")] = (shapes[k][0], 1)
    sd = _train_sd(shapes)
    x = rnd((B, 4, H, W, 96), 17, -0.5, 1.0).requires_grad_(True)
    g = rnd((B, 4, H, W, 96), 18, -1.0, 2.0)
    O.TRAIN = O.TrainCtx()
    try:
        with torch.enable_grad():
            y = O.ms_block(x, sd, "", 3, (2, 9, 9), tuple(shift), O.NeuronCfg(kind, 0.1, None, 2.0, 4))
            y.backward(g)
        running = dict(O.TRAIN.running)
    finally:
        O.TRAIN = None
    _close(y, TB[f"{tag}_y"], 1e-3, "y")
    _close(x.grad, TB[f"{tag}_gx"], 1e-3, "gx")
    for k in TB.files:
        if k.startswith(tag + "_g/"):
            name = k[len(tag) + 3:]
            if sd[name].grad is None:
                assert float(np.abs(TB[k]).max()) == 0.0, name          # e.g. the dead attn_sn parameters
                continue
            if name.endswith("proj.bias"):          # a bias in front of a batch-stat BN: its gradient is rounding noise
                assert sd[name].grad.abs().max().item() < 1e-4 * float(np.abs(TB[f"{tag}_g/attn.proj.weight"]).mean())
                continue
            _close(sd[name].grad, TB[k], 2e-3, name)
        if k.startswith(tag + "_r/"):
            name = k[len(tag) + 3:]
            prefix, stat = name.rsplit("running_", 1)
            _close(running[prefix][0 if stat == "mean" else 1], TB[k], 1e-3, name)


def test_oracle_train_mode_patch_merging_matches_reference():
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    sd = _train_sd({"reduction.weight": (192, 384), "norm.norm_layer.weight": (192,), "norm.norm_layer.bias": (192,),
                    "norm.norm_layer.running_mean": (192,), "norm.norm_layer.running_var": (192,)})
    x = rnd((2, 4, 9, 21, 96), 19, -0.5, 1.0).requires_grad_(True)
    O.TRAIN = O.TrainCtx()
    try:
        with torch.enable_grad():
            y = O.ms_patch_merge(x, sd, "", O.NeuronCfg("lif", 0.1, None, 2.0, 4))
            y.backward(rnd(tuple(y.shape), 20, -1.0, 2.0))
    finally:
        O.TRAIN = None
    _close(y, TB["merge_y"], 1e-3, "y")
    _close(x.grad, TB["merge_gx"], 1e-3, "gx")
    _close(sd["reduction.weight"].grad, TB["merge_g/reduction.weight"], 1e-3, "dW")


def test_oracle_train_step_matches_reference():
    """Whole-model TRAIN-mode forward + loss + backward (3-encoder model, 144 x 144, batch 2): the loss and the gradient
    norm of every parameter against the reference.  The net is chaotic, so single spike flips
    between two BLAS code paths move individual norms: the bar is the loss to 2 % and 90 % of the norms to 20 %."""
    from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel, synth_label
    import yaml
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "sdformerflow_amd", "configs",
                                           "train_DSEC_supervised_SDformerFlow_en4.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type="lif")
    cfg["swin_transformer"].update(input_size=[144, 144], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    model = MS_SpikingformerFlowNet(cfg["model"].copy(), cfg["swin_transformer"].copy())
    sd = _train_sd({k: tuple(v.shape) for k, v in model.state_dict().items()})
    chunk = O.prepare_chunk(synth_voxel(2, 10, 144, 144, seed=1234 + 4))
    label, mask = synth_label(2, 144, 144)
    ocfg = {"neuron": O.NeuronCfg("lif", 0.1, None, 2.0, 10), "num_bins": 10, "window_size": (2, 9, 9),
            "depths": [2, 2, 6], "num_heads": [3, 6, 12]}
    O.TRAIN = O.TrainCtx()
    try:
        with torch.enable_grad():
            flows = O.forward_flownet(chunk, sd, ocfg)
            loss = O.flow_loss_supervised(flows, label, mask, 1.0, 1.0)
            loss.backward()
    finally:
        O.TRAIN = None
    assert abs(loss.item() - float(TS["loss"])) <= 0.02 * float(TS["loss"]), (loss.item(), float(TS["loss"]))
    names, ref = [str(n) for n in TS["grad_names"]], TS["grad_norms"]
    ok = tot = 0
    for n, r in zip(names, ref):
        g = sd[n].grad
        if r < 0:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        tot += 1
        ok += abs(float(g.norm()) - r) <= 0.2 * r + 1e-12
    assert ok >= 0.9 * tot, (ok, tot)


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_oracle_train_mode_is_pinned_exactly_by_the_spike_forced_reference_run(kind):
    """`train_step_forced.npz` (tests/golden/make_golden.py `gold_train_step_forced`, needs /root/reference) is the record of the
    oracle's TRAIN-mode forward + backward run with the REAL reference's spikes forced into every neuron layer: both graphs then
    carry identical spike trains, so unlike the free-running statistic above the comparison is exact - loss to 1e-7, every
    parameter gradient to 5e-5 of its largest element, and every decision in which the oracle's own pre-activation would have
    differed from the reference's explained by a 16-ulp threshold margin (0 unexplained).  The same forced run against the GPU's
    spikes is tests/test_train_gpu.py::test_whole_model_train_step_spike_forced_gradient_parity."""
    F_ = np.load(os.path.join(os.path.dirname(__file__), "golden", "train_step_forced.npz"))
    lr, lo = (float(v) for v in F_[f"{kind}_loss"])
    assert abs(lr - lo) <= 1e-7 * abs(lr)
    fun = F_[f"{kind}_flips_unexplained_n"]
    assert len(fun) == 78 and int(fun[:, 1].sum()) == 0                         # 78 forced neuron layers, nothing unexplained
    assert int(fun[:, 0].sum()) <= 1e-6 * int(fun[:, 2].sum())                  # a handful of 188 M decisions sit on the threshold
    rel = F_[f"{kind}_grad_rel"]
    live = rel[rel >= 0]
    assert len(live) >= 200 and float(live.max()) <= 5e-5, float(live.max())


@pytest.mark.parametrize("tag,kind", [("w15_sw", "lif"), ("w15_w", "psn"), ("t20_sw", "lif")])
def test_ms_block_config5_flavours_match_reference(tag, kind):
    """BASELINE configs[4]: the large window (2,15,15) (450-token positional encoding, shift (1,7,7), padding 33 -> 45) and
    D = T = 20 - the oracle's window / shift index arithmetic and neuron scan beyond the (2,9,9), T = 4 fixtures."""
    g = gold("ms_block_config5")
    D, H, W, w0, w1, w2, s0, s1, s2 = [int(v) for v in g[f"{tag}_cfg"]]
    C, nH = 96, 3
    shapes = block_shapes(C, nH, kind, D)
    shapes["attn.positional_encoding"] = (1, nH, w0 * w1 * w2, C // nH)
    sd = synth_state_dict(shapes)
    x = rnd((1, D, H, W, C), 23, -0.5, 1.0)
    y = O.ms_block(x, sd, "", nH, (w0, w1, w2), (s0, s1, s2), ncfg(kind, D))
    ref = g[f"{tag}_y"]
    bad = np.abs(y.numpy() - ref) > 1e-4 * np.abs(ref).mean()
    assert bad.mean() <= 1e-4, bad.mean()           # a handful of spike flips at most (addmm vs the fixed fmaf order for psn)


# ------------------------------------------------------------------ SEW model family (SpikingformerFlowNet)
def _schema_shapes(path):
    shapes = {}
    for line in open(path):
        parts = line.split()
        shapes[parts[0]] = tuple(int(v) for v in parts[1].split("x")) if len(parts) > 1 and parts[1] else ()
    return shapes


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_sew_flownet_end_to_end_matches_reference(kind):
    """`forward_sew_flownet` against the REAL reference's `SpikingformerFlowNet` (fixture sew_end_to_end, 3 encoders, 144 x 192):
    flow maps bit for bit, the firing rate of all 75 neuron layers in call order, and the AEE tuple."""
    g = gold("sew_end_to_end")
    here = os.path.join(os.path.dirname(__file__), "golden")
    shapes = {k: v for k, v in _schema_shapes(os.path.join(here, f"state_schema_sew_{kind}.txt")).items()
              if not k.endswith(("relative_position_index", "relative_coords_table", "num_batches_tracked"))}
    sd = synth_state_dict(shapes)
    ocfg = {"neuron": O.NeuronCfg(kind, 0.1, None, 2.0, 10), "num_bins": 10, "window_size": (2, 9, 9), "depths": [2, 2, 6],
            "num_heads": [3, 6, 12]}
    chunk = O.prepare_chunk(synth_voxel(1, 10, 144, 192, seed=1234 + 7))
    O.RATE_LOG, O.PSN_MODE = [], "addmm"                                   # the reference's literal BLAS call pins end-to-end
    try:
        with torch.no_grad():
            flows = O.forward_sew_flownet(chunk, sd, ocfg)
        rates = list(O.RATE_LOG)
    finally:
        O.RATE_LOG, O.PSN_MODE = None, "fmaf"
    for i, f in enumerate(flows):
        s = f.shape[-1] // (24 * 2 ** i)
        assert torch.equal(f[:, :, ::s, ::s], torch.from_numpy(g[f"{kind}_flow{i}"])), i
    assert len(rates) == len(g[f"{kind}_rates"]) == 75
    for (p, r), name, want in zip(rates, g[f"{kind}_rate_names"], g[f"{kind}_rates"]):
        assert p.rstrip(".") == str(name) and abs(r - float(want)) < 1e-7, (p, name, r, want)
    label, mask = synth_label(1, 144, 192)
    assert np.allclose(np.array([float(v.reshape(-1)[0]) for v in O.aee(flows[-1], label, mask, 1.0)]), g[f"{kind}_aee"], rtol=1e-6)


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_sew_module_tree_has_the_reference_schema_and_flops(kind):
    """`SpikingformerFlowNet` builds on CPU with the reference's exact state_dict (names, order, shapes) and its analytic
    `flops()` / `record_flops()` reproduce the reference's numbers (8 481 322 080 / 89 entries summing to 8 464 476 672 at 144 x 192)."""
    import yaml
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import SpikingformerFlowNet
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind)
    cfg["swin_transformer"].update(input_size=[144, 192], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    m = SpikingformerFlowNet(cfg["model"].copy(), cfg["swin_transformer"].copy())
    want = _schema_shapes(os.path.join(os.path.dirname(__file__), "golden", f"state_schema_sew_{kind}.txt"))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert list(got) == list(want) and got == want
    assert m.flops() == 8481322080

    def flat(d):
        return sum((flat(v) if isinstance(v, dict) else [v] for v in d.values()), [])
    rec = flat(m.record_flops())
    assert len(rec) == 89 and sum(rec) == 8464476672
