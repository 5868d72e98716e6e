"""Deterministic synthetic inputs and weights.

Everything is generated from numpy's PCG64 seeded by a CRC32 of the tensor *name*, so the GPU box
regenerates bit-identical weights without shipping a 220 MB checkpoint.  Shapes come from the
module tree (or any name->shape mapping); distributions follow the reference's `init_weights`
(models/STSwinNet_SNN/Spiking_STSwinNet.py:264-276) where it defines one, and are deliberately
non-degenerate elsewhere (BN running stats, positional_encoding, PSN bias) so that spikes fire.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch


def _rng(name: str, salt: int = 0) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) + 7919 * salt) & 0xFFFFFFFF))


def synth_tensor(name: str, shape, salt: int = 0, psn_bias: float = -0.1) -> torch.Tensor:
    """Value for state_dict entry `name` of `shape`."""
    g = _rng(name, salt)
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 1)[0] if "." in name else ""
    n = int(np.prod(shape)) if shape else 1
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        a = g.normal(0.0, 0.1, shape)
    elif leaf == "running_var":
        a = g.uniform(0.5, 1.5, shape)
    elif leaf == "positional_encoding":
        a = g.normal(0.0, 0.3, shape)
    elif leaf == "relative_position_bias_table":
        a = g.normal(0.0, 0.5, shape)
    elif leaf == "logit_scale":
        a = np.log(10.0) + g.normal(0.0, 0.1, shape)
    elif parent.endswith("spiking_neuron"):                      # PSN (T,T) weight / (T,1) bias
        if leaf == "weight":
            T = shape[0]
            a = g.uniform(-1.0, 1.0, shape) / np.sqrt(T)
            a = a + np.eye(T) * 0.5
        else:
            a = np.full(shape, psn_bias)
    elif leaf == "weight" and len(shape) == 1:                    # BN / LN scale
        a = g.uniform(0.8, 1.2, shape)
    elif leaf == "bias" and (".norm" in name or ".bn" in name or "_bn." in name):
        a = g.normal(0.0, 0.05, shape)
    elif leaf == "bias":
        a = g.normal(0.0, 0.02, shape)
    elif leaf == "weight" and len(shape) == 2:                    # Linear: kaiming normal, fan_out
        a = g.normal(0.0, np.sqrt(2.0 / shape[0]), shape)
    elif leaf == "weight" and len(shape) == 4:                    # conv / deconv: xavier uniform
        rf = shape[2] * shape[3]
        bound = np.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
        a = g.uniform(-bound, bound, shape)
    else:
        a = g.normal(0.0, 0.02, shape)
    return torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))


def synth_state_dict(shapes: dict, salt: int = 0, psn_bias: float = -0.1) -> dict:
    """name -> tensor for every entry of `shapes` (name -> shape)."""
    return {k: synth_tensor(k, s, salt, psn_bias) for k, s in shapes.items()}


def synth_voxel(B: int, bins: int, H: int, W: int, seed: int = 1234, density: float = 0.10) -> torch.Tensor:
    """DSEC-like signed event voxel (B,bins,H,W): `density` of pixels per bin active,
    values ~ U(-1,1)*(1+Exp(0.5)) clipped to +-4, else exactly 0."""
    g = np.random.Generator(np.random.PCG64(seed))
    active = g.random((B, bins, H, W)) < density
    val = g.uniform(-1.0, 1.0, (B, bins, H, W)) * (1.0 + g.exponential(0.5, (B, bins, H, W)))
    return torch.from_numpy(np.where(active, np.clip(val, -4.0, 4.0), 0.0).astype(np.float32))


def synth_label(B: int, H: int, W: int, seed: int = 4321):
    """(label ~ N(0,5^2) px (B,2,H,W), mask ~ Bernoulli(0.7) (B,1,H,W))."""
    g = np.random.Generator(np.random.PCG64(seed))
    label = g.normal(0.0, 5.0, (B, 2, H, W)).astype(np.float32)
    mask = (g.random((B, 1, H, W)) < 0.7).astype(np.float32)
    return torch.from_numpy(label), torch.from_numpy(mask)


def synth_uniform(shape, seed: int, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """Seeded U(lo,hi) fp32 tensor (test inputs; shared by the golden generator and the tests)."""
    g = np.random.Generator(np.random.PCG64(seed))
    return torch.from_numpy(g.uniform(lo, hi, shape).astype(np.float32))
