"""MI355X-native SDformerFlow forward hot path.  `install_reference_aliases()` makes this package answer to the
reference's import names."""
import sys
import types


def install_reference_aliases():
    """Register this package's modules under the names the reference's scripts import (eval_DSEC_flow_SNN.py:1-16,
    train_flow_parallel_supervised_SNN.py): `models.STSwinNet_SNN.*`, `models.STSwinNet.*`, `configs.parser`,
    `loss.flow_supervised`, `DSEC_dataloader.DSEC_dataset_lite`, `utils.utils.load_model` and
    `spikingjelly.activation_based.{functional, neuron}` (neuron.LIFNode / IFNode: the holders `set_backend` filters on).
    Idempotent; never shadows a real `spikingjelly` that is already imported."""
    from . import STSwinNet, STSwinNet_SNN, DSEC_dataloader, checkpoint, configs, loss, spikingjelly_compat
    from .STSwinNet import PatchEmbed, STSwinNet as ann_net, load_pretrained, swin_transformer3D_v2
    from .STSwinNet_SNN import Spiking_modules, Spiking_STSwinNet, Spiking_submodules, Spiking_swin_transformer3D
    from .DSEC_dataloader import DSEC_dataset_lite
    from .configs import parser
    from .loss import flow_supervised
    models = types.ModuleType("models")
    models.STSwinNet_SNN, models.STSwinNet = STSwinNet_SNN, STSwinNet
    utils = types.ModuleType("utils")
    utils_utils = types.ModuleType("utils.utils")
    utils_utils.load_model = checkpoint.load_model
    utils.utils = utils_utils
    table = {
        "models": models, "models.STSwinNet_SNN": STSwinNet_SNN, "models.STSwinNet": STSwinNet,
        "models.STSwinNet_SNN.Spiking_STSwinNet": Spiking_STSwinNet, "models.STSwinNet_SNN.Spiking_modules": Spiking_modules,
        "models.STSwinNet_SNN.Spiking_submodules": Spiking_submodules,
        "models.STSwinNet_SNN.Spiking_swin_transformer3D": Spiking_swin_transformer3D,
        "models.STSwinNet.STSwinNet": ann_net, "models.STSwinNet.PatchEmbed": PatchEmbed,
        "models.STSwinNet.swin_transformer3D_v2": swin_transformer3D_v2, "models.STSwinNet.load_pretrained": load_pretrained,
        "configs": configs, "configs.parser": parser, "loss": loss, "loss.flow_supervised": flow_supervised,
        "DSEC_dataloader": DSEC_dataloader, "DSEC_dataloader.DSEC_dataset_lite": DSEC_dataset_lite,
        "utils": utils, "utils.utils": utils_utils,
    }
    if "spikingjelly" not in sys.modules:
        sj = types.ModuleType("spikingjelly")
        ab = types.ModuleType("spikingjelly.activation_based")
        neuron = types.ModuleType("spikingjelly.activation_based.neuron")
        neuron.LIFNode, neuron.IFNode = Spiking_submodules.LIFNode, Spiking_submodules.IFNode
        ab.functional, ab.neuron, sj.activation_based = spikingjelly_compat.functional, neuron, ab
        table.update({"spikingjelly": sj, "spikingjelly.activation_based": ab, "spikingjelly.activation_based.neuron": neuron})
    for name, mod in table.items():
        sys.modules.setdefault(name, mod)
    return sorted(table)
