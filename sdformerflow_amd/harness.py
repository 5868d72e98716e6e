"""The model-facing steps of the reference's evaluation loop (eval_DSEC_flow_SNN.valid_test :153-271),
restated so that the harness logic runs with this package: input preparation, forward, metric accumulation.
Dataset loading, MLflow and visualisation are out of scope."""
import torch
import torch.nn.functional as F


def center_crop(t, size):
    """Centre crop of the last two dims (reference DSEC_dataloader/data_augmentation.py:62-86)."""
    H, W = t.shape[-2:]
    th, tw = size
    i, j = (H - th) // 2, (W - tw) // 2
    return t[..., i:i + th, j:j + tw]


def prepare_chunk(voxel, norm_input="minmax", spike_th=None, polarity=True):
    """signed voxel (B,bins,H,W) -> network input (B,bins,2,H,W).

    pos/neg split stacked on dim 2 (eval_DSEC_flow_SNN.py:179-186), min-max over the non-zeros of the
    WHOLE batch tensor (:199-205) or mean/std (:206-212), optional binarisation (:215-217)."""
    chunk = torch.stack((F.relu(voxel), F.relu(-voxel)), dim=2) if polarity else voxel
    nz = chunk != 0
    if nz.any():
        vals = chunk[nz]
        if norm_input == "minmax":
            lo, hi = vals.min(), vals.max()
            if lo != hi:
                chunk = torch.where(nz, (chunk - lo) / (hi - lo), chunk)
        elif norm_input == "std":
            mean, std = vals.mean(), vals.std()
            if std > 0:
                chunk = torch.where(nz, (chunk - mean) / std, chunk)
    if spike_th is not None:
        chunk = torch.where(chunk > spike_th, torch.ones_like(chunk), torch.where(chunk < spike_th, torch.zeros_like(chunk), chunk))
    return chunk


def evaluate(model, samples, config, device="cuda"):
    """Run `model` over an iterable of (chunk (B,bins,H,W), mask (B,H,W), label (B,2,H,W)) like
    valid_test does and return the running-mean metrics dict (AEE, PE1-3, outliers) (:253-271, :283-305)."""
    from .loss.flow_supervised import AEE
    from .spikingjelly_compat import functional
    crop = config["loader"].get("crop")
    tot = {"AEE": 0.0, "PE1": 0.0, "PE2": 0.0, "PE3": 0.0, "outliers": 0.0}
    it = 0
    for chunk, mask, label in samples:
        functional.reset_net(model)
        chunk, label = chunk.to(device, torch.float32), label.to(device, torch.float32)
        mask = mask.to(device).unsqueeze(1).float()
        if crop:
            chunk, label, mask = (center_crop(t, crop) for t in (chunk, label, mask))
        x = prepare_chunk(chunk, config["model"].get("norm_input"), config["data"].get("spike_th"),
                          config["loader"].get("polarity", True))
        with torch.no_grad():
            pred = model(x)["flow"][-1]
        if config["metrics"].get("mask_events"):
            mask = mask * x.sum(1).sum(1, keepdim=True).bool()
        m = AEE(pred, label, mask, config["metrics"]["flow_scaling"])()
        for b in range(pred.shape[0]):
            it += 1
            tot["AEE"] += float(m[0][b])
            for key, v in zip(("PE1", "PE2", "PE3", "outliers"), m[1:]):
                tot[key] += float(v.reshape(-1)[b] if v.numel() > 1 else v)
    return {k: v / max(it, 1) for k, v in tot.items()}
