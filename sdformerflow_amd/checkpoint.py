"""Checkpoint loading with the reference's semantics, minus MLflow.

`utils/utils.py:10-40` (`load_model`) of the reference resolves an MLflow run id to `<artifact>/model/data/model.pth`,
a pickle of the WHOLE module, takes its `state_dict()`, optionally strips DataParallel's `module.` prefix (test=True),
optionally remaps position tensors, and loads with `strict=False`.  A whole-module pickle can only be opened where the
reference's classes and spikingjelly are importable, so the on-disk format here is the plain `state_dict` with the
reference's keys (one `torch.save(model.state_dict(), path)` in the reference environment converts a run).
Everything after the unpickling is the same: prefix strip, remap, `strict=False`, "no model found" leaves the model as is.
"""
import os

import torch

from .STSwinNet.load_pretrained import load_pretrained_interpolate


def read_state_dict(path, device="cpu"):
    """A `.pth` holding a state_dict (or {"state_dict": ...} / {"model": ...}); tensors only (weights_only)."""
    obj = torch.load(path, map_location=device, weights_only=True)
    for key in ("state_dict", "model"):
        if isinstance(obj, dict) and key in obj and isinstance(obj[key], dict):
            obj = obj[key]
    if not isinstance(obj, dict) or not all(torch.is_tensor(v) for v in obj.values()):
        raise ValueError(f"{path}: not a state_dict")
    return dict(obj)


def load_model(path, model, device, remap=None, test=False):
    """Reference `load_model(prev_runid, model, device, remap, test)` with the run id replaced by the file it resolves to.
    Returns the model; a missing file leaves it untouched (the reference's behaviour for an unknown run)."""
    if not path or not os.path.isfile(path):
        print(f"No model found at {path}\n")
        return model
    pretrained = read_state_dict(path, device)
    if test:
        pretrained = {k.replace("module.", ""): v for k, v in pretrained.items()}
    if remap == "v2":
        raise NotImplementedError('remap "v2" needs scipy.interpolate.interp2d (removed from SciPy >= 1.14); see load_pretrained.py')
    if remap == "v1":
        load_pretrained_interpolate(model, pretrained)
    model.load_state_dict(pretrained, strict=False)
    print(f"Model restored from {path}\n")
    return model


def save_state_dict_file(model, path):
    """Plain state_dict with the reference's keys (what `load_model` reads)."""
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, path)
