#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference is imported behind `oracle/stubs` (spikingjelly / timm stand-ins, see
oracle/stubs/README.md).  No reference source or bytecode is written anywhere: the fixtures
hold inputs' *seeds*, small inputs, and the reference's OUTPUTS.  Weights are regenerated from
`sdformerflow_amd.synthetic` (name-keyed PCG64), so they are not stored either.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle", "stubs"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel, synth_label  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)

from spikingjelly.activation_based import functional, neuron  # noqa: E402
from models.STSwinNet_SNN import Spiking_swin_transformer3D as ref_swin  # noqa: E402
from models.STSwinNet_SNN.Spiking_submodules import PSN  # noqa: E402
from models.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet_en4  # noqa: E402
from models.STSwinNet import swin_transformer3D_v2 as ref_ann  # noqa: E402
from loss.flow_supervised import AEE  # noqa: E402
from configs.parser import YAMLParser  # noqa: E402


from sdformerflow_amd.synthetic import synth_uniform as rnd  # noqa: E402


def load_synth(module, salt=0, psn_bias=-0.1):
    # derived index/coordinate buffers keep their constructor values
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()
              if not k.endswith(("relative_position_index", "relative_coords_table"))}
    module.load_state_dict(synth_state_dict(shapes, salt, psn_bias), strict=False)
    functional.reset_net(module)
    functional.set_step_mode(module, "m")
    module.eval()
    return module


def spk_kwargs(kind, T):
    return {"num_steps": T, "v_reset": None, "v_th": 0.1, "neuron_type": kind,
            "surrogate_fun": "surrogate.ATan()", "tau": 2.0, "detach_reset": True, "spike_norm": "BN"}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


# ------------------------------------------------------------------ 1. neurons
def gold_neurons():
    out = {}
    for T in (2, 10, 20):
        x = rnd((T, 4096), 100 + T, -0.3, 0.6)
        x[:, :64] = 0.1                      # exactly at threshold: h = 0.05, 0.1-ish boundary cases
        x[0, 64:128] = 0.2                   # h == v_th exactly at t = 0  (0 + 0.2/2 = 0.1)
        for tag, v_reset in (("soft", None), ("hard", 0.0)):
            n = neuron.LIFNode(tau=2.0, v_threshold=0.1, v_reset=v_reset, detach_reset=True, step_mode="m")
            s = n(x)
            out[f"lif_{tag}_T{T}_s"] = s.to(torch.uint8)
            out[f"lif_{tag}_T{T}_v"] = n.v
        p = PSN(T)
        sd = synth_state_dict({"spiking_neuron.weight": (T, T), "spiking_neuron.bias": (T, 1)}, salt=T)
        p.weight.copy_(sd["spiking_neuron.weight"])
        p.bias.copy_(sd["spiking_neuron.bias"])
        h = torch.addmm(p.bias, p.weight, x.flatten(1))
        out[f"psn_T{T}_w"], out[f"psn_T{T}_b"] = p.weight.data.clone(), p.bias.data.clone()
        out[f"psn_T{T}_h"] = h.to(torch.float32)
        out[f"psn_T{T}_s"] = p(x).to(torch.uint8)
    save("neurons", **out)


def gold_neurons_extra():
    """The three neuron types no shipped configuration uses (reference Spiking_modules.py:49-56, 75-92), through the reference's own
    `Spiking_neuron` switch in multi-step mode: plif with a non-trivial w, SLTTlif, glif with random gates."""
    from models.STSwinNet_SNN.Spiking_modules import Spiking_neuron as RefSN
    out = {}
    for T in (4, 10):
        x = rnd((T, 3, 8, 6, 10), 300 + T, -0.3, 0.6)
        out[f"x_T{T}"] = x
        for tag, v_reset in (("soft", None), ("hard", 0.0), ("hard05", 0.05)):
            for kind in ("plif", "SLTTlif"):
                if kind == "SLTTlif" and tag == "hard05":
                    continue
                m = RefSN(num_steps=T, neuron_type=kind, v_th=0.1, v_reset=v_reset, surrogate_fun="surrogate.ATan()", tau=2.0,
                          detach_reset=True)
                if kind == "plif":
                    m.spiking_neuron.w.fill_(0.3 * T - 2.0)
                    out[f"plif_T{T}_w"] = m.spiking_neuron.w.data.clone()
                functional.set_step_mode(m, "m")
                functional.reset_net(m)
                m.eval()
                out[f"{kind}_{tag}_T{T}_s"] = m(x).to(torch.uint8)
        g = RefSN(num_steps=T, neuron_type="glif", surrogate_fun="surrogate.ATan()")
        sd = {k: rnd(tuple(v.shape), 700 + 16 * T + i, -1.0, 2.0) for i, (k, v) in enumerate(g.state_dict().items())}
        g.load_state_dict(sd)
        functional.reset_net(g)
        g.eval()
        for k, v in sd.items():
            out[f"glif_T{T}/{k}"] = v
        out[f"glif_T{T}_s"] = g(3.0 * x).to(torch.uint8)                           # gates ~ sigmoid(U(-1, 1)), threshold ~ 0.5
    save("neurons_extra", **out)


# ------------------------------------------------------------------ 2. window index maps
def gold_index_maps():
    out = {}
    for tag, (B, D, H, W), ws, shift in (("s0", (1, 10, 72, 96), (2, 9, 9), (1, 4, 4)),
                                         ("odd", (2, 4, 18, 21), (2, 9, 9), (1, 4, 4)),
                                         ("noshift", (2, 4, 18, 21), (2, 9, 9), (0, 0, 0))):
        idx = torch.arange(B * D * H * W, dtype=torch.float32).view(B, D, H, W, 1) + 1.0   # 0 = padding
        wsz, ssz = ref_ann.get_window_size((D, H, W), ws, shift)
        pad_d = (wsz[0] - D % wsz[0]) % wsz[0]
        pad_b = (wsz[1] - H % wsz[1]) % wsz[1]
        pad_r = (wsz[2] - W % wsz[2]) % wsz[2]
        x = torch.nn.functional.pad(idx, (0, 0, 0, pad_r, 0, pad_b, 0, pad_d))
        _, Dp, Hp, Wp, _ = x.shape
        if any(i > 0 for i in ssz):
            x = torch.roll(x, shifts=(-ssz[0], -ssz[1], -ssz[2]), dims=(1, 2, 3))
        win = ref_swin.window_partition_v2(x, wsz)                  # (Wd, B_, Wh, Ww, 1)
        out[f"{tag}_gather"] = (win.reshape(win.shape[0] * win.shape[1], -1).to(torch.int64) - 1).to(torch.int32)
        # inverse path: (B_, N, C) -> view(-1, *ws, C) -> window_reverse -> roll back -> crop
        y = win.reshape(win.shape[1], -1, 1).view(-1, *wsz, 1)
        y = ref_ann.window_reverse(y, wsz, B, Dp, Hp, Wp)
        if any(i > 0 for i in ssz):
            y = torch.roll(y, shifts=ssz, dims=(1, 2, 3))
        y = y[:, :D, :H, :W, :]
        out[f"{tag}_roundtrip_ok"] = np.array(bool(torch.equal(y, idx)))
        out[f"{tag}_shape"] = np.array([B, D, H, W, *ws, *shift])
        m = ref_swin.compute_mask(Dp, Hp, Wp, wsz, ssz, torch.device("cpu"))
        out[f"{tag}_mask_sum"] = np.array(float(m.sum()))
        if tag != "s0":
            out[f"{tag}_mask"] = m.to(torch.int8)
    save("index_maps", **out)


# ------------------------------------------------------------------ 3. a5 QK attention
def gold_qk_attention():
    out = {}
    for tag, (B_, C, nH), kind in (("c96_lif", (4, 96, 3), "lif"), ("c96_psn", (4, 96, 3), "psn"),
                                   ("c192_lif", (2, 192, 6), "lif"), ("c384_psn", (1, 384, 12), "psn")):
        m = ref_swin.Spiking_QK_WindowAttention3D(C, (2, 9, 9), (0, 0, 0), nH, norm="BN", **spk_kwargs(kind, 10))
        load_synth(m)
        x = rnd((2, B_, 9, 9, C), 7 + C, -0.5, 1.0)
        y, _ = m(x)                                                   # (B_, 162, C)
        out[f"{tag}_y"] = y
        out[f"{tag}_cfg"] = np.array([B_, C, nH, 7 + C])
    save("qk_attention", **out)


def gold_qk_attention_scores():
    """Second return value of `Spiking_QK_WindowAttention3D.forward` (`attn = self.attn_sn(x)`, :709-711) on the inputs of
    `gold_qk_attention` - what `log=True` is written to collect per stage (kept in its own file: qk_attention.npz is unchanged)."""
    out = {}
    for tag, (B_, C, nH), kind in (("c96_lif", (4, 96, 3), "lif"), ("c96_psn", (4, 96, 3), "psn"),
                                   ("c192_lif", (2, 192, 6), "lif"), ("c384_psn", (1, 384, 12), "psn")):
        m = ref_swin.Spiking_QK_WindowAttention3D(C, (2, 9, 9), (0, 0, 0), nH, norm="BN", **spk_kwargs(kind, 10))
        load_synth(m)
        x = rnd((2, B_, 9, 9, C), 7 + C, -0.5, 1.0)
        _, attn = m(x)                                                # (2, B_, 9, 9, C) spikes
        out[f"{tag}_attn"] = np.packbits(attn.to(torch.uint8).numpy())
        out[f"{tag}_shape"] = np.array(attn.shape)
        out[f"{tag}_rate"] = np.array(float(attn.mean()))
    save("qk_attention_scores", **out)


# ------------------------------------------------------------------ 4. a9 SEW attention with mask
def gold_sew_attention():
    out = {}
    for tag, kind in (("lif", "lif"), ("psn", "psn")):
        C, nH, B, nW = 96, 3, 2, 4
        m = ref_swin.Spiking_BN_WindowAttention3D(C, (2, 9, 9), (0, 0, 0), nH, version="swinv1", norm="BN",
                                                  **spk_kwargs(kind, 10))
        load_synth(m)
        x = (rnd((2, B * nW, 9, 9, C), 11) > 0.4).float()           # spike input (SEW blocks feed spikes)
        mask = ref_swin.compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4), torch.device("cpu"))
        y, attn = m(x, mask)
        out[f"{tag}_y"] = y.to(torch.uint8)
        out[f"{tag}_attn_sum"] = attn.sum((-1, -2))
        out[f"{tag}_attn00"] = attn[0, 0]
        y2, _ = (functional.reset_net(m), m(x, None))[1]
        out[f"{tag}_y_nomask"] = y2.to(torch.uint8)
    save("sew_attention", **out)


# ------------------------------------------------------------------ 5. a10 ANN window attention with mask
def gold_ann_attention():
    out = {}
    C, nH, B, nW = 96, 3, 1, 4
    m = ref_ann.WindowAttention3D(C, (2, 9, 9), (0, 0, 0), nH, qkv_bias=True)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if "relative" not in k}
    sd = synth_state_dict(shapes)
    m.load_state_dict(sd, strict=False)
    m.eval()
    x = rnd((B * nW, 162, C), 13, -1.0, 1.0)
    mask = ref_ann.compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4), torch.device("cpu"))
    y, attn = m(x, mask)
    out["y"], out["attn00"] = y, attn[0, 0]
    y2, _ = m(x, None)
    out["y_nomask"] = y2
    out["rel_index_sum"] = np.array(int(m.relative_position_index.sum()))
    out["coords_table"] = m.relative_coords_table
    save("ann_attention", **out)


# ------------------------------------------------------------------ 6. full MS block, merge
def gold_ms_block():
    out = {}
    C, nH = 96, 3
    for tag, kind, (H, W), shift in (("lif_sw", "lif", (18, 21), (1, 4, 4)), ("lif_w", "lif", (9, 21), (0, 0, 0)),
                                     ("psn_sw", "psn", (9, 21), (1, 4, 4))):
        kw = spk_kwargs(kind, 4)
        blk = ref_swin.MS_Spiking_SwinTransformerBlock3D(C, (H, W), nH, window_size=(2, 9, 9),
                                                         shift_size=shift, norm_layer="BN", **kw)
        load_synth(blk)
        x = rnd((1, 4, H, W, C), 17, -0.5, 1.0)
        Hp, Wp = -(-H // 9) * 9, -(-W // 9) * 9
        wsz, ssz = ref_ann.get_window_size((4, H, W), (2, 9, 9), shift)
        mask = ref_swin.compute_mask(4, Hp, Wp, wsz, ssz, torch.device("cpu"))
        out[f"{tag}_y"] = blk(x, mask)
        out[f"{tag}_cfg"] = np.array([H, W, *shift])
    for kind in ("lif", "psn"):
        pm = ref_swin.MS_SpikingPatchMerging((9, 21), C, norm_layer="BN", **spk_kwargs(kind, 4))
        load_synth(pm)
        out[f"{kind}_merge_y"] = pm(rnd((1, 4, 9, 21, C), 19, -0.5, 1.0))
    save("ms_block", **out)


# ------------------------------------------------------------------ 7. end to end
def en4_config(kind):
    cfg = YAMLParser("/root/reference/configs/train_DSEC_supervised_SDformerFlow_en4.yml")
    config = cfg.combine_entries(cfg.config)
    config["swin_transformer"]["input_size"] = [288, 384]
    config["model"]["spiking_neuron"]["neuron_type"] = kind
    return config


def gold_end_to_end():
    out = {}
    for kind in ("lif", "psn"):
        config = en4_config(kind)
        model = MS_SpikingformerFlowNet_en4(config["model"].copy(), config["swin_transformer"].copy())
        load_synth(model)
        rates = []
        hooks = []
        for name, mod in model.named_modules():
            if name.endswith("spiking_neuron"):
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, name=name: rates.append((name, float(o.mean())))))
        vox = synth_voxel(1, 10, 288, 384, seed=1234 + 1)
        # harness prep (eval_DSEC_flow_SNN.py:179-205), reference semantics, in place
        chunk = torch.cat((torch.relu(vox).unsqueeze(2), torch.relu(-vox).unsqueeze(2)), dim=2)
        lo, hi = chunk[chunk != 0].min(), chunk[chunk != 0].max()
        chunk[chunk != 0] = (chunk[chunk != 0] - lo) / (hi - lo)
        functional.reset_net(model)
        res = model(chunk)
        for h in hooks:
            h.remove()
        preds = model.sttmultires_unet(chunk) if False else None
        for i, f in enumerate(res["flow"]):
            s = f.shape[-1] // (24 * 2 ** i)                     # native resolution of prediction i
            out[f"{kind}_flow{i}"] = f[:, :, ::s, ::s].contiguous()
            out[f"{kind}_flow{i}_abs_mean"] = np.array(float(f.abs().mean()))
        label, mask = synth_label(1, 288, 384)
        m = AEE(res["flow"][-1], label, mask, 1)()
        out[f"{kind}_aee"] = np.array([float(v.reshape(-1)[0]) for v in m])
        out[f"{kind}_rates"] = np.array([r for _, r in rates], dtype=np.float32)
        out[f"{kind}_rate_names"] = np.array([n for n, _ in rates])
        out[f"{kind}_n_state"] = np.array(len(model.state_dict()))
        out[f"{kind}_n_param"] = np.array(sum(p.numel() for p in model.parameters()))
        out[f"{kind}_chunk_checksum"] = np.array(float(chunk.double().sum()))
    save("end_to_end", **out)
    # state_dict schema (names + shapes only) for the drop-in boundary test
    for kind in ("lif", "psn"):
        config = en4_config(kind)
        model = MS_SpikingformerFlowNet_en4(config["model"].copy(), config["swin_transformer"].copy())
        with open(os.path.join(HERE, f"state_schema_en4_{kind}.txt"), "w") as f:
            for k, v in model.state_dict().items():
                f.write(f"{k} {'x'.join(map(str, v.shape))}\n")


# ------------------------------------------------------------------ 7b. SEW model family end to end
def gold_sew_end_to_end():
    """`SpikingformerFlowNet` (SEW shortcuts, 3 encoders) at 144 x 192, lif and psn: flow maps at native resolution, the AEE
    tuple, the firing rate of every neuron layer in call order, and the state_dict schema."""
    from models.STSwinNet_SNN.Spiking_STSwinNet import SpikingformerFlowNet
    out = {}
    for kind in ("lif", "psn"):
        config = en4_config(kind)
        config["swin_transformer"].update(input_size=[144, 192], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
        model = SpikingformerFlowNet(config["model"].copy(), config["swin_transformer"].copy())
        load_synth(model)
        rates, hooks = [], []
        for name, mod in model.named_modules():
            if name.endswith("spiking_neuron"):
                hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: rates.append((name, float(o.mean())))))
        vox = synth_voxel(1, 10, 144, 192, seed=1234 + 7)
        chunk = torch.cat((torch.relu(vox).unsqueeze(2), torch.relu(-vox).unsqueeze(2)), dim=2)
        lo, hi = chunk[chunk != 0].min(), chunk[chunk != 0].max()
        chunk[chunk != 0] = (chunk[chunk != 0] - lo) / (hi - lo)
        functional.reset_net(model)
        res = model(chunk)
        for h in hooks:
            h.remove()
        for i, f in enumerate(res["flow"]):
            s = f.shape[-1] // (24 * 2 ** i)
            out[f"{kind}_flow{i}"] = f[:, :, ::s, ::s].contiguous()
        label, mask = synth_label(1, 144, 192)
        m = AEE(res["flow"][-1], label, mask, 1)()
        out[f"{kind}_aee"] = np.array([float(v.reshape(-1)[0]) for v in m])
        out[f"{kind}_rates"] = np.array([r for _, r in rates], dtype=np.float32)
        out[f"{kind}_rate_names"] = np.array([n for n, _ in rates])
        with open(os.path.join(HERE, f"state_schema_sew_{kind}.txt"), "w") as f:
            for k, v in model.state_dict().items():
                f.write(f"{k} {'x'.join(map(str, v.shape))}\n")
    save("sew_end_to_end", **out)


# ------------------------------------------------------------------ 8. ANN STTFlowNet end to end (BASELINE config 3 family)
def gold_ann_end_to_end():
    from models.STSwinNet.STSwinNet import STTFlowNet, STTFlowNet_4en
    cfg = YAMLParser("/root/reference/configs/train_DSEC_supervised_STT_voxel.yml")
    config = cfg.combine_entries(cfg.config)
    config["swin_transformer"]["input_size"] = [144, 192]
    model = STTFlowNet(config["model"].copy(), config["swin_transformer"].copy())
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()
              if not k.endswith(("relative_position_index", "relative_coords_table"))}
    model.load_state_dict(synth_state_dict(shapes), strict=False)
    model.eval()
    x = synth_voxel(2, 20, 144, 192, seed=1234 + 3)
    res = model(x, None)
    out = {}
    for i, f in enumerate(res["flow"]):
        s = f.shape[-1] // (24 * 2 ** i)                       # predictions live at 18x24, 36x48, 72x96
        out[f"flow{i}"] = f[:, :, ::s, ::s].contiguous()
    out["n_state"] = np.array(len(model.state_dict()))
    save("ann_end_to_end", **out)
    with open(os.path.join(HERE, "state_schema_sttflownet.txt"), "w") as fh:
        for k, v in model.state_dict().items():
            fh.write(f"{k} {'x'.join(map(str, v.shape))}\n")


def gold_ann_odd_size():
    """STTFlowNet at 150 x 200: stage sizes 38 x 50, 19 x 25, 10 x 13 - every decoder output is one larger than its skip, which
    `skip_concat` (models/model_util.py:14-19) crops.  Flows subsampled 4 x (the finest prediction lives at 76 x 100 before upsampling)."""
    from models.STSwinNet.STSwinNet import STTFlowNet
    cfg = YAMLParser("/root/reference/configs/train_DSEC_supervised_STT_voxel.yml")
    config = cfg.combine_entries(cfg.config)
    config["swin_transformer"]["input_size"] = [150, 200]
    model = STTFlowNet(config["model"].copy(), config["swin_transformer"].copy())
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()
              if not k.endswith(("relative_position_index", "relative_coords_table"))}
    model.load_state_dict(synth_state_dict(shapes), strict=False)
    model.eval()
    res = model(synth_voxel(1, 20, 150, 200, seed=9), None)
    save("ann_odd_size", **{f"flow{i}": f[:, :, ::2, ::2].contiguous() for i, f in enumerate(res["flow"])})


def gold_neuron_grads():
    """Gradients through the reference's own `Spiking_neuron` factory (Spiking_modules.py:26-99): autograd over the
    spikingjelly-stub LIFNode (torch backend) and the in-tree PSN, ATan surrogate, seeded x and upstream gradient."""
    from models.STSwinNet_SNN.Spiking_modules import Spiking_neuron
    out = {}
    with torch.enable_grad():
        for T in (2, 10):
            x0 = rnd((T, 2048), 300 + T, -0.3, 0.6)
            x0[:, :64] = 0.1
            x0[0, 64:128] = 0.2                                   # h == v_th exactly at t = 0
            g = rnd((T, 2048), 400 + T, -1.0, 2.0)
            for tag, kw in (("soft_detach", dict(v_reset=None, detach_reset=True)), ("soft_nodetach", dict(v_reset=None, detach_reset=False)),
                            ("hard_detach", dict(v_reset=0.0, detach_reset=True)), ("hard_nodetach", dict(v_reset=0.0, detach_reset=False)),
                            ("tau3_soft_detach", dict(v_reset=None, detach_reset=True, tau=3.0))):
                kw = dict(dict(tau=2.0), **kw)
                n = Spiking_neuron(num_steps=T, neuron_type="lif", v_th=0.1, surrogate_fun="surrogate.ATan()", **kw)
                functional.set_step_mode(n, "m")
                x = x0.clone().requires_grad_(True)
                s = n(x)
                s.backward(g)
                out[f"lif_{tag}_T{T}_s"] = s.detach().to(torch.uint8)
                out[f"lif_{tag}_T{T}_gx"] = x.grad.clone()
            n = Spiking_neuron(num_steps=T, neuron_type="psn", v_th=0.1, surrogate_fun="surrogate.ATan()")
            sd = synth_state_dict({"spiking_neuron.weight": (T, T), "spiking_neuron.bias": (T, 1)}, salt=T)
            with torch.no_grad():
                n.spiking_neuron.weight.copy_(sd["spiking_neuron.weight"])
                n.spiking_neuron.bias.copy_(sd["spiking_neuron.bias"])
            x = x0.clone().requires_grad_(True)
            s = n(x)
            s.backward(g)
            out[f"psn_T{T}_s"] = s.detach().to(torch.uint8)
            out[f"psn_T{T}_gx"] = x.grad.clone()
            out[f"psn_T{T}_gW"] = n.spiking_neuron.weight.grad.clone()
            out[f"psn_T{T}_gb"] = n.spiking_neuron.bias.grad.clone()
    save("neuron_grads", **out)


def _no_drop_path(model):
    """DropPath (timm, stochastic depth 0.2 hard-wired at Spiking_STSwinNet.py:65) draws from the global RNG: replaced by
    the identity for the fixtures so that they are a function of the seeded inputs only."""
    for m in model.modules():
        if hasattr(m, "drop_path"):
            m.drop_path = torch.nn.Identity()


def gold_train_block():
    """TRAIN-mode forward + backward of the reference's MS swin block and patch merging (batch-stat BN, ATan surrogate,
    detached reset) on seeded inputs: output, dL/dx, parameter gradients, BN running-stat updates."""
    out = {}
    C, nH = 96, 3
    with torch.enable_grad():
        for tag, kind, (B, H, W), shift in (("lif_sw", "lif", (2, 18, 21), (1, 4, 4)), ("psn_w", "psn", (1, 9, 21), (0, 0, 0))):
            kw = spk_kwargs(kind, 4)
            blk = ref_swin.MS_Spiking_SwinTransformerBlock3D(C, (H, W), nH, window_size=(2, 9, 9), shift_size=shift,
                                                             norm_layer="BN", **kw)
            load_synth(blk)
            blk.train()
            _no_drop_path(blk)
            functional.reset_net(blk)
            x = rnd((B, 4, H, W, C), 17, -0.5, 1.0).requires_grad_(True)
            g = rnd((B, 4, H, W, C), 18, -1.0, 2.0)
            Hp, Wp = -(-H // 9) * 9, -(-W // 9) * 9
            wsz, ssz = ref_ann.get_window_size((4, H, W), (2, 9, 9), shift)
            mask = ref_swin.compute_mask(4, Hp, Wp, wsz, ssz, torch.device("cpu"))
            y = blk(x, mask)
            y.backward(g)
            out[f"{tag}_y"], out[f"{tag}_gx"] = y.detach(), x.grad.clone()
            out[f"{tag}_cfg"] = np.array([B, H, W, *shift])
            for n, prm in blk.named_parameters():
                if prm.grad is not None:
                    out[f"{tag}_g/{n}"] = prm.grad.clone()
            for n, buf in blk.named_buffers():
                if n.endswith(("running_mean", "running_var")):
                    out[f"{tag}_r/{n}"] = buf.clone()
        pm = ref_swin.MS_SpikingPatchMerging((9, 21), C, norm_layer="BN", **spk_kwargs("lif", 4))
        load_synth(pm)
        pm.train()
        functional.reset_net(pm)
        x = rnd((2, 4, 9, 21, C), 19, -0.5, 1.0).requires_grad_(True)
        y = pm(x)
        g = rnd(tuple(y.shape), 20, -1.0, 2.0)
        y.backward(g)
        out["merge_y"], out["merge_gx"] = y.detach(), x.grad.clone()
        out["merge_g/reduction.weight"] = pm.reduction.weight.grad.clone()
    save("train_block", **out)


def gold_train_step():
    """One TRAIN-mode forward + backward of the whole reference model (3-encoder MS_SpikingformerFlowNet, 144 x 144, batch 2,
    lif) with the reference's loss (`flow_loss_supervised`, gamma None): loss, flows, gradient norm of every parameter."""
    from models.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet
    from loss.flow_supervised import flow_loss_supervised
    config = en4_config("lif")
    config["swin_transformer"].update(input_size=[144, 144], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    out = {}
    with torch.enable_grad():
        model = MS_SpikingformerFlowNet(config["model"].copy(), config["swin_transformer"].copy())
        load_synth(model)
        model.train()
        _no_drop_path(model)
        functional.reset_net(model)
        vox = synth_voxel(2, 10, 144, 144, seed=1234 + 4)
        chunk = torch.cat((torch.relu(vox).unsqueeze(2), torch.relu(-vox).unsqueeze(2)), dim=2)
        lo, hi = chunk[chunk != 0].min(), chunk[chunk != 0].max()
        chunk[chunk != 0] = (chunk[chunk != 0] - lo) / (hi - lo)
        label, mask = synth_label(2, 144, 144)
        res = model(chunk)
        loss = flow_loss_supervised(config, "cpu")(res["flow"], label, mask, gamma=None)
        loss.backward()
        out["loss"] = np.array(float(loss))
        for i, f in enumerate(res["flow"]):
            out[f"flow{i}_abs_mean"] = np.array(float(f.abs().mean()))
        names, norms = [], []
        for n, prm in model.named_parameters():
            names.append(n)
            norms.append(float(prm.grad.norm()) if prm.grad is not None else -1.0)
        out["grad_names"], out["grad_norms"] = np.array(names), np.array(norms, dtype=np.float64)
        out["g/preds.2.conv.0.weight"] = model.sttmultires_unet.preds[2].conv[0].weight.grad.clone()
        out["g/preds.2.conv.0.bias"] = model.sttmultires_unet.preds[2].conv[0].bias.grad.clone()
    save("train_step", **out)


def gold_train_step_forced():
    """The oracle's TRAIN mode pinned EXACTLY on the reference (the free-running comparison of `train_step` is only a statistic:
    the net is chaotic).  The reference's train-mode forward + backward runs with forward hooks that keep every neuron layer's
    spikes; the oracle then runs its own train-mode forward + backward with those spikes FORCED (`oracle.NEURON_FORCE`: value of
    the spike = the reference's, gradient = the surrogate's at the oracle's membrane, reset following the forced decision), so the
    two compute graphs carry identical spike trains and differ by floating-point rounding only.  Stored: per parameter
    max|g_oracle - g_reference| / max|g_reference|, the loss pair, and per neuron layer the number of the reference's decisions the
    oracle's own pre-activation would have taken differently (`delta_consistent`: all within 16 ulp of the threshold).  lif and psn."""
    from models.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet
    from loss.flow_supervised import flow_loss_supervised
    from oracle import sdformer_oracle as O
    out = {}
    for kind in ("lif", "psn"):
        config = en4_config(kind)
        config["swin_transformer"].update(input_size=[144, 144], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
        with torch.enable_grad():
            model = MS_SpikingformerFlowNet(config["model"].copy(), config["swin_transformer"].copy())
            load_synth(model)
            model.train()
            _no_drop_path(model)
            functional.reset_net(model)
            tape = {}
            hooks = [m.register_forward_hook(lambda mod, inp, o, n=n: tape.__setitem__(n + ".", o.detach().clone()))
                     for n, m in model.named_modules() if n.endswith(".spiking_neuron")]
            vox = synth_voxel(2, 10, 144, 144, seed=1234 + 4)
            chunk = torch.cat((torch.relu(vox).unsqueeze(2), torch.relu(-vox).unsqueeze(2)), dim=2)
            lo, hi = chunk[chunk != 0].min(), chunk[chunk != 0].max()
            chunk[chunk != 0] = (chunk[chunk != 0] - lo) / (hi - lo)
            label, mask = synth_label(2, 144, 144)
            res = model(chunk)
            loss = flow_loss_supervised(config, "cpu")(res["flow"], label, mask, gamma=None)
            loss.backward()
            for h in hooks:
                h.remove()
            ref_grads = {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()}
            # ---- the oracle on the same weights / inputs with the reference's spikes forced
            sd0 = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.endswith("num_batches_tracked")}
        # (running statistics were updated in place by the reference's forward: the oracle needs the values BEFORE it - regenerate)
        shapes = {k: tuple(v.shape) for k, v in sd0.items()}
        sd = synth_state_dict(shapes, 0, -0.1)
        sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v.clone())
              for k, v in sd.items()}
        ocfg = {"neuron": O.NeuronCfg(kind, 0.1, None, 2.0, 10), "num_bins": 10, "window_size": (2, 9, 9), "depths": [2, 2, 6],
                "num_heads": [3, 6, 12]}
        report, used = [], set()

        def force(prefix, x, kind=kind, sd=sd):
            got = tape.get(prefix)
            if got is None:
                return None                                             # the integer-input token gate: never taped, computed freely
            used.add(prefix)
            xd = x.detach()
            delta = 16 * 2.0 ** -23 * max(float(xd.pow(2).mean().sqrt()), 0.1)
            r = O.delta_consistent(xd, got.reshape(xd.shape), ocfg["neuron"], {k: v.detach() for k, v in sd.items()}, prefix, delta)
            report.append((prefix, r["flips"], r["unexplained"], r["n"]))
            return got.reshape(x.shape)

        O.TRAIN, O.NEURON_FORCE = O.TrainCtx(), force
        try:
            with torch.enable_grad():
                flows = O.forward_flownet(chunk, sd, ocfg)
                oloss = O.flow_loss_supervised(flows, label, mask, 1.0, 1.0)
                oloss.backward()
        finally:
            O.TRAIN, O.NEURON_FORCE = None, None
        names, rel = [], []
        for n, g in ref_grads.items():
            og = sd[n].grad
            names.append(n)
            if g is None or float(g.abs().max()) == 0.0:
                assert og is None or float(og.abs().max()) == 0.0, n
                rel.append(-1.0)
            elif n.endswith("attn.proj.bias"):
                # a bias in front of a batch-statistics BatchNorm: its true gradient is zero, both sides hold rounding noise -
                # measured against the scale of the same layer's weight gradient
                rel.append(float((og - g).abs().max() / ref_grads[n[:-4] + "weight"].abs().max()))
            else:
                rel.append(float((og - g).abs().max() / g.abs().max()))
        worst = max(rel)
        print(f"  train_step_forced[{kind}]: loss reference {float(loss):.8f} oracle {float(oloss):.8f}; {len(used)} layers forced, "
              f"{sum(r[1] for r in report)} decisions differ, {sum(r[2] for r in report)} unexplained of {sum(r[3] for r in report)}; "
              f"worst parameter-gradient deviation {worst:.2e} ({names[int(np.argmax(rel))]})")
        out[f"{kind}_loss"] = np.array([float(loss), float(oloss)])
        out[f"{kind}_grad_names"], out[f"{kind}_grad_rel"] = np.array(names), np.array(rel, dtype=np.float64)
        out[f"{kind}_layers"] = np.array([r[0] for r in report])
        out[f"{kind}_flips_unexplained_n"] = np.array([[r[1], r[2], r[3]] for r in report], dtype=np.int64)
    save("train_step_forced", **out)


def gold_ms_block_config5():
    """The config-5 flavours of the block (BASELINE configs[4]): D = 20 frames with T = 20 neurons, and the large window
    (2,15,15) with its 450-token positional encoding and shift (1,7,7) - pins the oracle's index arithmetic beyond (2,9,9)."""
    out = {}
    C, nH = 96, 3
    for tag, kind, ws, (D, H, W), shift in (("w15_sw", "lif", (2, 15, 15), (6, 15, 33), (1, 7, 7)),
                                            ("w15_w", "psn", (2, 15, 15), (4, 30, 15), (0, 0, 0)),
                                            ("t20_sw", "lif", (2, 9, 9), (20, 9, 12), (1, 4, 4))):
        blk = ref_swin.MS_Spiking_SwinTransformerBlock3D(C, (H, W), nH, window_size=ws, shift_size=shift, norm_layer="BN",
                                                         **spk_kwargs(kind, D))
        load_synth(blk)
        x = rnd((1, D, H, W, C), 23, -0.5, 1.0)
        Hp, Wp = -(-H // ws[1]) * ws[1], -(-W // ws[2]) * ws[2]
        wsz, ssz = ref_ann.get_window_size((D, H, W), ws, shift)
        mask = ref_swin.compute_mask(D, Hp, Wp, wsz, ssz, torch.device("cpu"))
        out[f"{tag}_y"] = blk(x, mask)
        out[f"{tag}_cfg"] = np.array([D, H, W, *ws, *shift])
    save("ms_block_config5", **out)


def gold_formats():
    """On-disk formats: the reference's `load_pretrained_interpolate` on seeded position tensors
    (window 9 -> 15, i.e. the large-window variant of BASELINE config 5) and its `DSECDatasetLite` read-back of the tiny
    tree of tests/golden/dsec_tree.py.  h5py / torchvision are imported by the reference's dataset module but not used
    on this path: empty module objects stand in for them here (generation time only)."""
    import tempfile
    import types
    from models.STSwinNet.load_pretrained import load_pretrained_interpolate as ref_interp
    src = {"blk.attn.relative_position_bias_table": rnd((3 * 17 * 17, 3), 11, -1.0, 2.0),
           "blk.attn.positional_encoding": rnd((1, 3, 162, 32), 12, -1.0, 2.0),
           "absolute_pos_embed": rnd((1, 16, 8), 13, -1.0, 2.0),
           "blk.attn.relative_position_index": torch.zeros(162, 162),
           "blk.attn.relative_coords_table": torch.zeros(1, 3, 17, 17, 3),
           "blk.attn_mask": torch.zeros(4, 162, 162),
           "blk.mlp.fc1.weight": rnd((8, 4), 14, -1.0, 2.0)}

    class Target(torch.nn.Module):
        def state_dict(self, *a, **k):
            return {"blk.attn.relative_position_bias_table": torch.zeros(3 * 29 * 29, 3),
                    "blk.attn.positional_encoding": torch.zeros(1, 3, 450, 32),
                    "absolute_pos_embed": torch.zeros(1, 36, 8), "blk.mlp.fc1.weight": torch.zeros(8, 4)}
    sd = {k: v.clone() for k, v in src.items()}
    ref_interp(Target(), sd)
    out = {"interp_keys": np.array(sorted(sd.keys()))}
    for k, v in sd.items():
        out["interp/" + k] = v
    for name in ("h5py", "torchvision", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    from DSEC_dataloader.DSEC_dataset_lite import DSECDatasetLite as RefDS
    from dsec_tree import make_tree, config
    with tempfile.TemporaryDirectory() as root:
        make_tree(root)
        for tag, kw in (("vox1", {}), ("vox2", {"num_chunks": 2}), ("pol1", {"polarity": False}),
                        ("cnt2", {"encoding": "cnt", "num_chunks": 2}), ("list1", {"encoding": "list", "preprocessed": False})):
            ds = RefDS(config(root, **kw), "train")
            out[f"ds/{tag}/len"] = np.array(len(ds))
            for i in (0, len(ds) - 1):
                chunk, mask, label = ds[i]
                if isinstance(chunk, dict):
                    for kk, vv in chunk.items():
                        out[f"ds/{tag}/{i}/chunk_{kk}"] = vv
                else:
                    out[f"ds/{tag}/{i}/chunk"] = chunk
                out[f"ds/{tag}/{i}/mask"] = mask
                out[f"ds/{tag}/{i}/label"] = label
    save("formats", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["neurons", "neurons_extra", "index_maps", "qk_attention", "sew_attention", "ann_attention",
                             "qk_attention_scores", "ms_block", "end_to_end", "sew_end_to_end", "ann_end_to_end", "ann_odd_size", "formats", "neuron_grads", "train_block", "train_step", "train_step_forced", "ms_block_config5"]
    for w in which:
        globals()["gold_" + w]()


