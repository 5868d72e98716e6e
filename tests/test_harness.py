"""The product's evaluation harness and metric class against the reference's own numbers, driven through the REFERENCE's
import names (`sdformerflow_amd.install_reference_aliases()`).

CPU part: `loss.flow_supervised.AEE` on the reference's flow map (fixture end_to_end.npz) reproduces the reference's AEE /
PE1-3 / outlier tuple; the alias recipe resolves every class the reference's eval script imports.
GPU part: a fixture-fed `harness.evaluate` loop (the restated eval_DSEC_flow_SNN.valid_test :153-271) through the aliased
names on the real model."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = np.load(os.path.join(ROOT, "tests", "golden", "end_to_end.npz"))


def ref_flow(kind):
    """The reference's final flow map: stored at its native resolution, nearest-upsampled x2 by the model (:291-302)."""
    return torch.from_numpy(G[f"{kind}_flow3"]).repeat_interleave(2, -1).repeat_interleave(2, -2)


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_metric_class_reproduces_the_reference_tuple(kind):
    from sdformerflow_amd.loss.flow_supervised import AEE
    from sdformerflow_amd.synthetic import synth_label
    label, mask = synth_label(1, 288, 384)
    m = AEE(ref_flow(kind), label, mask, 1)()
    got = np.array([float(v.reshape(-1)[0]) for v in m])
    assert np.allclose(got, G[f"{kind}_aee"], rtol=1e-6, atol=1e-9), (got, G[f"{kind}_aee"])


def test_alias_recipe_resolves_the_reference_import_lines():
    """In a fresh interpreter (sys.modules untouched): the import lines of eval_DSEC_flow_SNN.py:4-6,10,14,16 work after the
    one call INTEGRATION.md prescribes, and give this package's classes."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import sdformerflow_amd; sdformerflow_amd.install_reference_aliases()\n"
        "from configs.parser import YAMLParser\n"
        "from loss.flow_supervised import *\n"
        "from models.STSwinNet_SNN.Spiking_STSwinNet import SpikingformerFlowNet, MS_SpikingformerFlowNet, MS_SpikingformerFlowNet_en4\n"
        "from models.STSwinNet.STSwinNet import STTFlowNet, STTFlowNet_4en\n"
        "from utils.utils import load_model\n"
        "from DSEC_dataloader.DSEC_dataset_lite import DSECDatasetLite\n"
        "from spikingjelly.activation_based import functional, neuron\n"
        "from models.STSwinNet_SNN.Spiking_submodules import *\n"
        "assert AEE.__module__ == 'sdformerflow_amd.loss.flow_supervised' and PSN.__module__.startswith('sdformerflow_amd')\n"
        "assert getattr(neuron, 'LIFNode') is LIFNode and callable(functional.reset_net)\n"
        "print('aliases ok')\n") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "aliases ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_evaluate_loop_through_reference_names_on_the_gpu(kind):
    """valid_test's model-facing steps (:153-271) with the reference's names: YAML -> ctor kwargs, load_state_dict,
    reset_net / set_step_mode / set_backend tree walks, `harness.evaluate` over fixture-fed samples.  The AEE tuple is the
    statistic of a free-running chaotic forward against a random label: within 2e-3 of the
    reference's; its exactness is the business of tests/test_replay_gpu.py.  What IS exact here: evaluate's numbers equal
    the metric class applied to the model's own flow, and a two-sample loop averages per sample like :262-271."""
    import sdformerflow_amd
    sdformerflow_amd.install_reference_aliases()
    net = importlib.import_module("models.STSwinNet_SNN.Spiking_STSwinNet")
    fs = importlib.import_module("loss.flow_supervised")
    functional = importlib.import_module("spikingjelly.activation_based").functional
    neuron = importlib.import_module("spikingjelly.activation_based").neuron
    sub = importlib.import_module("models.STSwinNet_SNN.Spiking_submodules")
    from sdformerflow_amd import harness
    from sdformerflow_amd.synthetic import synth_label, synth_state_dict, synth_voxel
    cfg = yaml.safe_load(open(os.path.join(ROOT, "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind)
    cfg["swin_transformer"]["input_size"] = [288, 384]
    cfg.setdefault("loader", {})["crop"] = None
    cfg["metrics"] = {"mask_events": False, "flow_scaling": 1}
    model = net.MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy())
    model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}), strict=True)
    model = model.to("cuda:0").eval()
    functional.reset_net(model)
    functional.set_step_mode(model, cfg["data"]["step_mode"])
    functional.set_backend(model, "cupy", sub.PSN if kind == "psn" else neuron.LIFNode)
    vox = synth_voxel(1, 10, 288, 384, seed=1234 + 1)                 # the fixture's voxel
    label, mask = synth_label(1, 288, 384)
    res = harness.evaluate(model, [(vox, mask, label)], cfg, device="cuda:0")
    ref = G[f"{kind}_aee"]
    print(kind, "evaluate:", res, "reference tuple:", ref)
    assert abs(res["AEE"] - ref[0]) <= 2e-3 * ref[0]
    for key, r in zip(("PE1", "PE2", "PE3", "outliers"), ref[1:]):
        assert abs(res[key] - r) <= 5e-3, (key, res[key], r)
    # exact: evaluate == metric class on the model's own flow (same prep, same mask handling)
    with torch.no_grad():
        flow = model(harness.prepare_chunk(vox).to("cuda:0"))["flow"][-1]
    m = fs.AEE(flow, label.to("cuda:0"), mask.to("cuda:0").unsqueeze(1).float(), 1)()
    assert abs(res["AEE"] - float(m[0][0])) <= 1e-6 * ref[0] and abs(res["PE3"] - float(m[3])) <= 1e-7
    two = harness.evaluate(model, [(vox, mask, label), (synth_voxel(1, 10, 288, 384, seed=99), mask, label)], cfg, device="cuda:0")
    one_b = harness.evaluate(model, [(synth_voxel(1, 10, 288, 384, seed=99), mask, label)], cfg, device="cuda:0")
    assert abs(two["AEE"] - 0.5 * (res["AEE"] + one_b["AEE"])) <= 1e-6 * ref[0]
