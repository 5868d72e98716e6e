"""Checkpoint key surgery for window-size / resolution changes.

Mirrors `models/STSwinNet/load_pretrained.py:91-178` (`load_pretrained_interpolate`) of the reference: the
derived buffers are dropped (they are rebuilt by the constructor) and the three learned position tensors
are resampled to the shape the *current* model expects:

  * `relative_position_bias_table`  ((2Wd-1)(2Wh-1)(2Ww-1), nH) with Wd = 2  -> bicubic over (Sh, Sw), the 3 temporal
                                     offsets kept as channels                                (reference :113-132)
  * `absolute_pos_embed`            (1, S*S, C)                 -> bicubic over (S, S)        (reference :134-155)
  * `positional_encoding`           (1, nH, 2*S*S, hd)          -> trilinear over (2, S, S)   (reference :158-178)
    (the learnable PE of Spiking_QK_WindowAttention3D; temporal window fixed at 2)

The reference's other remap (`remap_pretrained_keys_swin`, "v2") needs `scipy.interpolate.interp2d`, which SciPy
removed in 1.14 (this image: newer) - it cannot run in the reference's own code on this image either and no shipped
config selects it; `load_model(remap="v2")` raises.
"""
import torch
import torch.nn.functional as F

_DERIVED = ("relative_position_index", "relative_coords_table", "attn_mask")


def load_pretrained_interpolate(model, state_dict):
    """In place on `state_dict` (as the reference); returns it for convenience."""
    for k in [k for k in state_dict if any(t in k for t in _DERIVED)]:
        del state_dict[k]
    current = model.state_dict()
    for k in [k for k in state_dict if "relative_position_bias_table" in k]:
        old, new = state_dict[k], current[k]
        (L1, nH1), (L2, nH2) = old.shape, new.shape
        if nH1 != nH2:
            print(f"Error in loading {k}, passing......")
        elif L1 != L2:
            S1, S2 = int((L1 / 3) ** 0.5), int((L2 / 3) ** 0.5)
            r = F.interpolate(old.permute(1, 0).reshape(nH1, 3, S1, S1), size=(S2, S2), mode="bicubic")
            state_dict[k] = r.reshape(nH2, L2).permute(1, 0)
    for k in [k for k in state_dict if "absolute_pos_embed" in k]:
        old, new = state_dict[k], current[k]
        (_, L1, C1), (_, L2, _) = old.shape, new.shape
        if L1 != L2:
            S1, S2 = int(L1 ** 0.5), int(L2 ** 0.5)
            r = F.interpolate(old.reshape(-1, S1, S1, C1).permute(0, 3, 1, 2), size=(S2, S2), mode="bicubic")
            state_dict[k] = r.permute(0, 2, 3, 1).flatten(1, 2)
    for k in [k for k in state_dict if "positional_encoding" in k]:
        old, new = state_dict[k], current[k]
        (B, nH1, L1, C1), (_, nH2, L2, _) = old.shape, new.shape
        if nH1 != nH2:
            print(f"Error in loading {k}, passing......")
        elif L1 != L2:
            S1, S2 = int((L1 / 2) ** 0.5), int((L2 / 2) ** 0.5)
            r = F.interpolate(old.permute(0, 1, 3, 2).reshape(nH1, C1, 2, S1, S1), size=(2, S2, S2), mode="trilinear")
            state_dict[k] = r.reshape(B, nH1, C1, L2).permute(0, 1, 3, 2)
    return state_dict
