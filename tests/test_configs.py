"""The configs/*.yml surface of the drop-in boundary: every shipped training config, parsed with the
reference's parser semantics, constructs its named model through the reference's own idiom
(`eval(config["model"]["name"])(config["model"].copy(), config["swin_transformer"].copy())`, eval_DSEC_flow_SNN.py:87-90)
- and on the GPU that model runs a forward at the config's crop size."""
import os

import pytest
import torch

from sdformerflow_amd.configs.parser import YAMLParser
from sdformerflow_amd.STSwinNet.STSwinNet import *            # noqa: F401,F403  (the reference star-imports the model modules)
from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import *  # noqa: F401,F403
from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel

CFG_DIR = os.path.join(os.path.dirname(__file__), "..", "sdformerflow_amd", "configs")
TRAIN = ["train_DSEC_supervised_SDformerFlow_en4.yml", "train_DSEC_supervised_STT_voxel.yml",
         "train_MDR_supervised_SDformerFlow.yml", "train_MDR_supervised_STT_voxel.yml"]


def build(name):
    config = YAMLParser(os.path.join(CFG_DIR, name)).config
    config = YAMLParser.combine_entries(config)
    config["swin_transformer"]["input_size"] = list(config["loader"]["crop"])
    model = eval(config["model"]["name"])(config["model"].copy(), config["swin_transformer"].copy())
    return config, model


@pytest.mark.parametrize("name", TRAIN)
def test_shipped_config_builds_its_model(name):
    config, model = build(name)
    assert type(model).__name__ == config["model"]["name"]
    n = sum(p.numel() for p in model.parameters())
    assert n > 1e6
    with pytest.raises(Exception):                      # no CPU path: CPU tensors are refused, never silently computed
        model.eval()
        bins = config["model"]["num_bins"]
        H, W = config["loader"]["crop"]
        x = torch.zeros(1, bins, 2, H, W) if "Spiking" in config["model"]["name"] else torch.zeros(1, bins, H, W)
        model(x) if "Spiking" in config["model"]["name"] else model(x, None)


def test_eval_configs_parse():
    for name in ("valid_DSEC_supervised.yml", "eval_MV_supervised.yml"):
        cfg = YAMLParser(os.path.join(CFG_DIR, name)).config
        assert "metrics" in cfg and "loader" in cfg


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRAIN)
def test_shipped_config_forward_on_gpu(name):
    from sdformerflow_amd import harness
    config, model = build(name)
    skip = ("relative_position_index", "relative_coords_table", "num_batches_tracked")
    model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items() if not k.endswith(skip)}),
                          strict=False)
    model = model.eval().to("cuda:0")
    bins = config["model"]["num_bins"]
    H, W = config["loader"]["crop"]
    vox = synth_voxel(1, bins, H, W, seed=4242)
    with torch.no_grad():
        if "Spiking" in config["model"]["name"]:
            out = model(harness.prepare_chunk(vox).to("cuda:0"))
        else:
            out = model(vox.to("cuda:0"), None)
    flows = out["flow"]
    assert len(flows) == model.num_encoders and all(f.shape == (1, 2, H, W) and torch.isfinite(f).all() for f in flows)
