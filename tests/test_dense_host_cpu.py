"""Host side of the ANN path's dense kernels (no GPU): weight-plane packing and the slice plan of wide convolutions
(sdformerflow_amd/hip.py: pack_dense_conv_weight, pack_dense_linear_weight, dense_conv_slices, dense_conv_applicable)."""
import torch

from sdformerflow_amd import hip


def test_conv_weight_planes_reconstruct_the_weight_in_record_tap_channel_order():
    g = torch.Generator().manual_seed(1)
    w = torch.randn(32, 21, 3, 3, generator=g) * 0.05
    p = hip.pack_dense_conv_weight(w)
    assert p.shape == (2, 32, 2 * 144) and p.dtype == torch.float16
    sc = p.sdf_acc_scale                                                     # planes hold w / sc, sc a power of two
    assert 2.0 ** 14 <= float(w.abs().max()) / sc < 2.0 ** 15
    full = ((p[0].double() + p[1].double()) * sc).view(32, 2, 3, 3, 16)      # (n, record, ky, kx, channel in record)
    back = full.permute(0, 1, 4, 2, 3).reshape(32, 32, 3, 3)
    # 22 significant bits for EVERY weight down to 2^-13 of the largest (no fp16-subnormal lo halves: ADVICE r2)
    assert torch.all((back[:, :21] - w.double()).abs() <= w.double().abs() * 2.0 ** -21 + float(w.abs().max()) * 2.0 ** -38)
    assert torch.count_nonzero(back[:, 21:]) == 0                            # padding channels carry zero weights
    hi = (w / sc).half().float()                                             # plane 0 is the fp16 rounding of the scaled weight
    assert torch.equal(p[0].view(32, 2, 3, 3, 16).permute(0, 1, 4, 2, 3).reshape(32, 32, 3, 3)[:, :21].float(), hi)


def test_linear_weight_planes():
    g = torch.Generator().manual_seed(2)
    w = torch.randn(96, 64, generator=g)
    p = hip.pack_dense_linear_weight(w)
    assert p.shape == (2, 96, 64)
    sc = p.sdf_acc_scale
    assert torch.equal(p[0].float(), (w / sc).half().float())
    assert torch.all(((p[0].double() + p[1].double()) * sc - w.double()).abs() <= w.double().abs() * 2.0 ** -21 + float(w.abs().max()) * 2.0 ** -38)
    small = torch.full((96, 64), 1e-3)
    small[0, 0] = 0.5                                                        # a weight 500x below the largest keeps its bits
    q = hip.pack_dense_linear_weight(small)
    assert abs(float((q[0, 5, 5].double() + q[1, 5, 5].double()) * q.sdf_acc_scale) - 1e-3) <= 1e-3 * 2.0 ** -21


def test_slice_plan_and_applicability():
    assert hip.dense_conv_slices(6) == [(0, 6)]
    assert hip.dense_conv_slices(1) == [(0, 1)]
    assert hip.dense_conv_slices(13) == [(0, 6), (6, 6), (12, 1)]            # 194 channels: two sixes and the prediction record
    assert hip.dense_conv_slices(25) == [(0, 6), (6, 6), (12, 6), (18, 6), (24, 1)]
    assert hip.dense_conv_slices(48) == [(6 * i, 6) for i in range(8)]
    assert hip.dense_conv_slices(8) is None and hip.dense_conv_slices(3) is None
    assert hip.dense_conv_applicable(16, 288, 384, 96, 96) and hip.dense_conv_applicable(16, 288, 384, 10, 96)
    assert not hip.dense_conv_applicable(16, 288, 384, 32, 96)               # two records: no instantiation
    assert not hip.dense_conv_applicable(16, 288, 384, 96, 48)               # output columns come in blocks of 32
    assert not hip.dense_conv_applicable(64, 288, 384, 96, 96)               # 31-bit offsets
    assert hip.dense_linear_applicable(110592, 288, 96) and not hip.dense_linear_applicable(100, 64, 96)
    assert not hip.dense_linear_applicable(100, 96, 48)
