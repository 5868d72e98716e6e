"""Training path (BASELINE config 4) on the GPU: train-mode forward + backward of the product
(HIP neurons both ways, batch-stat BN, library GEMMs / convs) against fixtures produced by the REAL reference's
autograd (tests/golden/make_golden.py `train_block`, `train_step`) and against the CPU oracle's train mode.

Tolerances: an element counts as a mismatch when it is off by more than 1e-3 of the tensor's mean magnitude; the
allowed mismatch RATE is what the spike flips between two GEMM implementations (rocBLAS here, MKL in the fixture) cause
inside one block - the same teacher-forced logic as the forward parity tests."""
import os

import numpy as np
import pytest
import torch
import yaml

from sdformerflow_amd import train
from sdformerflow_amd.STSwinNet_SNN import Spiking_swin_transformer3D as SW
from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet
from sdformerflow_amd.synthetic import synth_label, synth_state_dict, synth_uniform as rnd, synth_voxel

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
HERE = os.path.dirname(os.path.abspath(__file__))
TB = np.load(os.path.join(HERE, "golden", "train_block.npz"))
TS = np.load(os.path.join(HERE, "golden", "train_step.npz"))
CFG = os.path.join(HERE, "..", "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")


def kw(kind, T):
    return {"num_steps": T, "v_reset": None, "v_th": 0.1, "neuron_type": kind, "surrogate_fun": "surrogate.ATan()", "tau": 2.0,
            "detach_reset": True, "spike_norm": "BN"}


def load_synth(mod):
    mod.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in mod.state_dict().items()}), strict=True)
    return mod.to(DEV).train()


def rate(got, ref):
    got, ref = got.detach().float().cpu(), torch.as_tensor(ref).float()
    assert got.shape == ref.shape
    scale = ref.abs().mean().item() + 1e-12
    return ((got - ref).abs() > 1e-3 * scale).float().mean().item()


@pytest.mark.parametrize("tag,kind", [("lif_sw", "lif"), ("psn_w", "psn")])
def test_train_mode_block_matches_reference_autograd(tag, kind):
    B, H, W, *shift = (int(v) for v in TB[f"{tag}_cfg"])
    blk = load_synth(SW.MS_Spiking_SwinTransformerBlock3D(96, (H, W), 3, window_size=(2, 9, 9), shift_size=tuple(shift),
                                                           norm_layer="BN", **kw(kind, 4)))
    x = rnd((B, 4, H, W, 96), 17, -0.5, 1.0).to(DEV).requires_grad_(True)
    g = rnd((B, 4, H, W, 96), 18, -1.0, 2.0).to(DEV)
    y = train.ms_block(x, blk, training=True)
    y.backward(g)
    report = {"y": rate(y, TB[f"{tag}_y"]), "gx": rate(x.grad, TB[f"{tag}_gx"])}
    params = dict(blk.named_parameters())
    for k in TB.files:
        if k.startswith(tag + "_g/"):
            name = k[len(tag) + 3:]
            if params[name].grad is None:
                assert float(np.abs(TB[k]).max()) == 0.0, name
            elif name.endswith("proj.bias"):               # a bias in front of a batch-stat BN: rounding noise on both sides
                assert params[name].grad.abs().max().item() < 1e-3 * float(np.abs(TB[f"{tag}_g/attn.proj.weight"]).mean())
            else:
                report[name] = rate(params[name].grad, TB[k])
        if k.startswith(tag + "_r/"):
            name = k[len(tag) + 3:]
            report["running:" + name] = rate(dict(blk.named_buffers())[name], TB[k])
    worst = max(report.values())
    print(f"train block {tag}: mismatch rates y {report['y']:.2e} gx {report['gx']:.2e} worst {worst:.2e} "
          f"({max(report, key=report.get)})")
    # measured on every box so far: 0 mismatching elements in y, gx and every parameter gradient; the bounds leave room for a
    # handful of spikes flipped by another box's library-GEMM heuristics (one flipped spike touches ~1e-4 of a tensor here)
    assert report["y"] <= 2e-3 and report["gx"] <= 5e-3 and worst <= 1e-2, report


def test_train_mode_patch_merging_matches_reference_autograd():
    pm = load_synth(SW.MS_SpikingPatchMerging((9, 21), 96, norm_layer="BN", **kw("lif", 4)))
    x = rnd((2, 4, 9, 21, 96), 19, -0.5, 1.0).to(DEV).requires_grad_(True)
    y = train.ms_patch_merge(x, pm)
    y.backward(rnd(tuple(y.shape), 20, -1.0, 2.0).to(DEV))
    assert rate(y, TB["merge_y"]) <= 1e-3 and rate(x.grad, TB["merge_gx"]) <= 1e-3
    assert rate(pm.reduction.weight.grad, TB["merge_g/reduction.weight"]) <= 1e-3


def small_kwargs(cfg, kind="lif"):
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind)
    cfg["swin_transformer"].update(input_size=[144, 144], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    return cfg["model"].copy(), cfg["swin_transformer"].copy()


def small_model(kind="lif"):
    model = load_synth(MS_SpikingformerFlowNet(*small_kwargs(yaml.safe_load(open(CFG)), kind)))
    for m in model.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0                                     # the fixture was made with DropPath = identity
    from sdformerflow_amd import harness
    chunk = harness.prepare_chunk(synth_voxel(2, 10, 144, 144, seed=1234 + 4)).to(DEV)
    label, mask = synth_label(2, 144, 144)
    return model, chunk, label.to(DEV), mask.to(DEV)


def test_whole_model_train_step_matches_reference():
    """Loss and per-parameter gradient norms of one train-mode forward + backward (3-encoder model, 144 x 144, batch 2)
    against the reference's (fixture `train_step`): loss within 2 %, 85 % of the gradient norms within 20 % (the net is
    chaotic, DESIGN.md section 2 - the CPU oracle meets the same bar against the same fixture)."""
    model, chunk, label, mask = small_model()
    flows = model(chunk)["flow"]
    assert len(flows) == 3 and all(f.shape == (2, 2, 144, 144) for f in flows)
    loss = train.flow_loss_supervised(flows, label, mask, 1.0, 1.0)
    loss.backward()
    ref_loss = float(TS["loss"])
    assert abs(loss.item() - ref_loss) <= 0.02 * ref_loss, (loss.item(), ref_loss)
    params = dict(model.named_parameters())
    ok = tot = 0
    for n, r in zip((str(n) for n in TS["grad_names"]), TS["grad_norms"]):
        g = params[n].grad
        if r < 0:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        tot += 1
        ok += abs(float(g.norm()) - r) <= 0.2 * r + 1e-12
    print(f"train step: loss {loss.item():.6f} vs reference {ref_loss:.6f}; {ok} of {tot} gradient norms within 20 %")
    assert ok >= 0.85 * tot, (ok, tot)          # measured 206-209 of 225 across builds / boxes (library GEMM heuristics differ)
    # last layer: 2 numbers, within 5 % of the LARGER one (as the forced-spike test states deviations: of the tensor's largest element).
    # The smaller is a cancelling sum an eighth of the larger; free-running trajectories move it by +-8 % of itself (library Linear
    # products: 0.105; csrc/linear_train.hip: 0.0997; reference 0.1073) - the exact statement is the spike-forced test below.
    gb = params["sttmultires_unet.preds.2.conv.0.bias"].grad.cpu()
    rb = torch.from_numpy(TS["g/preds.2.conv.0.bias"])
    assert float((gb - rb).abs().max()) <= 0.05 * float(rb.abs().max()), gb
    # ... and the hand-written Linear products may not sit farther from the reference than the library's products do on the same
    # step by more than 3 % of the larger element (a regression of the new kernels cannot hide behind the loosened bound)
    lib_linear = train.LINEAR_HIP
    try:
        train.LINEAR_HIP = False
        model2, chunk2, label2, mask2 = small_model()
        train.flow_loss_supervised(model2(chunk2)["flow"], label2, mask2, 1.0, 1.0).backward()
        gl = dict(model2.named_parameters())["sttmultires_unet.preds.2.conv.0.bias"].grad.cpu()
    finally:
        train.LINEAR_HIP = lib_linear
    assert float((gb - rb).abs().max()) <= float((gl - rb).abs().max()) + 0.03 * float(rb.abs().max()), (gb, gl, rb)


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_whole_model_train_step_spike_forced_gradient_parity(kind):
    """BASELINE configs[3], exact: the spike-forced TRAINING replay.  The GPU runs one train-mode forward + loss + backward (3-encoder
    model, 144 x 144, batch 2) with a forward hook on every neuron module keeping its spikes.  The CPU oracle - pinned on the real
    reference the same way (tests/golden/train_step_forced.npz) - then runs ITS train-mode forward + backward with those spikes
    forced (`oracle.NEURON_FORCE`: value = the GPU's spike, reset following it, gradient = the surrogate's at the oracle's own
    membrane).  Both graphs carry identical spike trains, so nothing chaotic is left between them:
      * every forced decision that the oracle's own pre-activation would have taken differently lies within 16 ulp of the
        threshold (`delta_consistent`: 0 unexplained);
      * the loss agrees to 1e-6 and EVERY parameter gradient to 2e-4 of its largest element (a bias in front of a batch-statistics
        BatchNorm has a zero true gradient: measured against its layer's weight gradient)."""
    model, chunk, label, mask = small_model(kind)
    ocfg = {"num_bins": 10, "window_size": (2, 9, 9), "depths": [2, 2, 6], "num_heads": [3, 6, 12]}
    worst, checked, layers = forced_step_parity(model, chunk, label, mask, kind, ocfg)
    # every taped layer was forced: all 78 neuron layers of the model except the 10 token gates (integer inputs: exact, not taped)
    assert layers == 68
    assert checked >= 200 and worst <= 2e-4, worst        # measured: lif 3.8e-6, psn 3.6e-5 (a PSN bias: a sum over 1e7 cancelling terms; 1.1e-4 with the library's Linear products)


def forced_step_parity(model, chunk, label, mask, kind, ocfg):
    """One train-mode forward + loss + backward on the GPU with every neuron's spikes kept, then the CPU oracle's with those spikes
    forced: asserts 0 unexplained decisions and the loss to 1e-6; returns (worst gradient deviation relative to the tensor's largest
    element, parameter gradients checked, neuron layers forced)."""
    from oracle import sdformer_oracle as O
    tape = {}
    hooks = [m.register_forward_hook(lambda mod, inp, o, n=n: tape.__setitem__(n + ".", o.detach().to(torch.uint8).cpu()))
             for n, m in model.named_modules() if n.endswith(".spiking_neuron")]
    flows = model(chunk)["flow"]
    loss = train.flow_loss_supervised(flows, label, mask, 1.0, 1.0)
    loss.backward()
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith("num_batches_tracked")}
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v.clone())
          for k, v in synth_state_dict(shapes).items()}                     # the weights BEFORE the step (running statistics moved)
    ncfg = O.NeuronCfg(kind, 0.1, None, 2.0, 10)
    ocfg = dict(ocfg, neuron=ncfg)
    report = []

    def force(prefix, x):
        got = tape.get(prefix)
        if got is None:
            return None                                                     # the integer-input token gate runs free (exact by construction)
        xd = x.detach()
        delta = 16 * 2.0 ** -23 * max(float(xd.pow(2).mean().sqrt()), 0.1)
        r = O.delta_consistent(xd, got.reshape(xd.shape).float(), ncfg, {k: v.detach() for k, v in sd.items()}, prefix, delta)
        report.append((prefix, r["flips"], r["unexplained"], r["n"]))
        return got.reshape(x.shape).float()

    O.TRAIN, O.NEURON_FORCE = O.TrainCtx(), force
    try:
        with torch.enable_grad():
            oflows = O.forward_flownet(chunk.cpu(), sd, ocfg)
            oloss = O.flow_loss_supervised(oflows, label.cpu(), mask.cpu(), 1.0, 1.0)
            oloss.backward()
    finally:
        O.TRAIN, O.NEURON_FORCE = None, None
    flips, unexplained, n = (sum(r[i] for r in report) for i in (1, 2, 3))
    assert len(report) == len(tape) and unexplained == 0, (len(report), len(tape), [r for r in report if r[2]][:5])
    assert flips <= 2e-6 * n, (flips, n)
    assert abs(loss.item() - oloss.item()) <= 1e-6 * abs(oloss.item()), (loss.item(), oloss.item())
    params = dict(model.named_parameters())
    worst, worst_name, checked = 0.0, "", 0
    for name, p in params.items():
        og = sd[name].grad
        if og is None or float(og.abs().max()) == 0.0:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name                 # dead parameters stay dead
            continue
        scale = sd[name[:-4] + "weight"].grad.abs().max() if name.endswith("attn.proj.bias") else og.abs().max()
        dev = float((p.grad.cpu() - og).abs().max() / scale)
        checked += 1
        if dev > worst:
            worst, worst_name = dev, name
    print(f"train step, spikes forced ({kind}, {tuple(chunk.shape)}): {len(report)} neuron layers, {n} decisions, {flips} differ from the oracle's "
          f"own, 0 unexplained; loss {loss.item():.8f} vs {oloss.item():.8f}; {checked} parameter gradients, worst deviation {worst:.2e} of the "
          f"tensor's largest element ({worst_name})")
    return worst, checked, len(report)


def test_en4_train_step_at_full_size_spike_forced_gradient_parity():
    """The same statement for the model and size BASELINE configs[3] names: MS_SpikingformerFlowNet_en4 at 288 x 384,
    local batch 1 (the CPU oracle's autograd graph of a larger batch does not fit a test) - the training kernels of round 5
    (LinearHipFunction, LinearDwFunction, Conv3x3HipFunction, the batch-statistics BatchNorm) inside the whole model at its own shapes:
    0 unexplained of ~5e8 forced decisions, loss to 1e-6, every parameter gradient within 2e-4 of its largest element."""
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet_en4
    from sdformerflow_amd import harness
    cfg = yaml.safe_load(open(CFG))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type="lif")
    cfg["swin_transformer"]["input_size"] = [288, 384]
    model = load_synth(MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy()))
    for m in model.modules():
        if hasattr(m, "drop_path_rate"):
            m.drop_path_rate = 0.0
    chunk = harness.prepare_chunk(synth_voxel(1, 10, 288, 384, seed=1234 + 6)).to(DEV)
    label, mask = synth_label(1, 288, 384)
    ocfg = {"num_bins": 10, "window_size": (2, 9, 9), "depths": [2, 2, 6, 2], "num_heads": [3, 6, 12, 24]}
    worst, checked, layers = forced_step_parity(model, chunk, label.to(DEV), mask.to(DEV), "lif", ocfg)
    assert layers >= 80 and checked >= 250 and worst <= 2e-4, (layers, checked, worst)


def test_adamw_steps_reduce_the_loss_and_update_running_stats():
    """The loop body of train_flow_parallel_supervised_SNN.py (reset, forward, loss, backward, clip 100, AdamW 1e-4) on a
    fixed micro-batch, through flat gradient buckets: the loss goes down, running statistics move, eval still works."""
    model, chunk, label, mask = small_model()
    buckets = train.GradientBuckets(model.parameters())
    assert all(p.grad is not None and p.grad.data_ptr() >= b.data_ptr() for b in buckets.flat[:1] for p in buckets.buckets[0])
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)          # the reference's optimiser settings
    rm0 = model.sttmultires_unet.encoders.swin3d.patch_embed.head.norm_layer.norm_layer.running_mean.clone()
    w0 = model.sttmultires_unet.preds[2].conv[0].weight.detach().clone()
    losses = [train.train_step(model, opt, chunk, label, mask, buckets=buckets).item() for _ in range(10)]
    print("losses", ["%.4f" % v for v in losses])
    assert all(np.isfinite(losses)) and losses[-1] < 0.85 * losses[0] and losses[-1] < losses[-4]     # measured: 5.54 -> 3.78
    assert not torch.equal(w0, model.sttmultires_unet.preds[2].conv[0].weight.detach())
    assert not torch.equal(rm0, model.sttmultires_unet.encoders.swin3d.patch_embed.head.norm_layer.norm_layer.running_mean)
    model.eval()
    with torch.no_grad():
        out = model(chunk)["flow"]
    assert all(torch.isfinite(f).all() for f in out)


@pytest.mark.parametrize("planes", [2, 3])
def test_optional_spike_gemm_forward_of_linear_layers(planes):
    """train.SPIKE_LINEAR_PLANES (off by default): Linear on spikes through the inference spike GEMM under autograd - output
    within the weight-plane truncation of F.linear, gradients equal to those of F.linear on the same spikes."""
    from sdformerflow_amd.autograd import SpikeLinearFunction
    x = (torch.rand((2, 500, 96), device=DEV) < 0.3).float()
    lin = torch.nn.Linear(96, 192).to(DEV)
    g = torch.randn((2, 500, 192), device=DEV)
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, lin.weight, lin.bias)
    yr.backward(g)
    ref = (yr.detach(), xr.grad.clone(), lin.weight.grad.clone(), lin.bias.grad.clone())
    lin.zero_grad()
    xs = x.clone().requires_grad_(True)
    ys = SpikeLinearFunction.apply(xs, lin.weight, lin.bias, planes)
    ys.backward(g)
    assert (ys - ref[0]).abs().max().item() <= 2e-6 * ref[0].abs().max().item()
    for got, r in ((xs.grad, ref[1]), (lin.weight.grad, ref[2]), (lin.bias.grad, ref[3])):
        assert (got - r).abs().max().item() <= 1e-5 * r.abs().max().item()


def test_eval_after_training_uses_the_trained_weights_not_a_stale_engine():
    """eval -> train_step -> eval (the normal train / validate loop): the packed inference engine caches split weight planes,
    folded BN (alpha, beta) and PSN matrices; after a training step it must be rebuilt.  The second eval has to differ from
    the first and equal what a freshly built model with the same state_dict gives."""
    model, chunk, label, mask = small_model()
    model.eval()
    with torch.no_grad():
        before = [f.clone() for f in model(chunk)["flow"]]
    e0 = model.engine()
    assert model.engine() is e0                                              # unchanged weights: the plan is reused
    buckets = train.GradientBuckets(model.parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
    for _ in range(2):
        train.train_step(model, opt, chunk, label, mask, buckets=buckets)
    model.eval()
    with torch.no_grad():
        after = [f.clone() for f in model(chunk)["flow"]]
    assert model.engine() is not e0
    assert not any(torch.equal(a, b) for a, b in zip(before, after)), "validation ran on the pre-training weights"
    cfg = yaml.safe_load(open(CFG))
    fresh = type(model)(*small_kwargs(cfg))
    fresh.load_state_dict(model.state_dict(), strict=True)
    fresh = fresh.to(DEV).eval()
    with torch.no_grad():
        want = fresh(chunk)["flow"]
    assert all(torch.equal(a, b) for a, b in zip(after, want))
    # an in-place edit without train(): seen through the version stamp
    with torch.no_grad():
        model.sttmultires_unet.preds[2].conv[0].bias.add_(1.0)
        moved = model(chunk)["flow"]
    assert not torch.equal(moved[-1], after[-1])
