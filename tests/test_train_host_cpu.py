"""Host logic of the training path's round-5 kernels, no GPU: which layers go to csrc/linear_dw.hip / csrc/linear_train.hip, what CPU
tensors do (the library path: the gloo tests and the plumbing dry run use it - the HIP wrappers themselves refuse CPU tensors), and that
the ctypes descriptors have the layout include/sdformerflow_hip.h declares."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from sdformerflow_amd import hip, train


def test_which_shapes_the_weight_gradient_kernels_take():
    assert hip.linear_dw_applicable(276480, 96, 96) and hip.linear_dw_applicable(4320, 768, 3072)
    assert not hip.linear_dw_applicable(1000, 96, 80) and not hip.linear_dw_applicable(1000, 100, 96)        # 96-multiples only
    assert not hip.linear_dw_applicable(1 << 23, 96, 96)                                                       # 31-bit byte offsets
    assert hip.linear_train_applicable(285120, 384, 96) and not hip.linear_train_applicable(100, 96, 48)
    assert hip.conv3x3_dw_applicable(40, 96, 96, 144, 192) and hip.conv3x3_dw_applicable(40, 768, 768, 9, 12)
    assert not hip.conv3x3_dw_applicable(40, 48, 96, 144, 192)                                                 # the patch embedding's 48 -> 96 layer
    assert not hip.conv3x3_dw_applicable(400, 96, 96, 144, 192)                                                # too many pixels for 31-bit offsets


def test_cpu_tensors_take_the_library_path():
    lin = torch.nn.Linear(96, 192)
    x = (torch.rand(3, 7, 96) < 0.3).float()
    assert torch.equal(train._linear(x, lin), F.linear(x, lin.weight, lin.bias))
    conv = torch.nn.Conv2d(96, 96, 3, 1, 1, bias=False)
    s = (torch.rand(2, 1, 96, 6, 5) < 0.3).float()
    y = train._conv_seq(s, conv, spikes=True)
    assert torch.equal(y, F.conv2d(s.flatten(0, 1), conv.weight, None, 1, 1).view(2, 1, 96, 6, 5))
    with pytest.raises(hip.SdfError):
        hip.linear_dw(torch.zeros(4, 96), torch.zeros(4, 96))                 # the kernels themselves: device tensors only
    with pytest.raises(hip.SdfError):
        hip.linear_train(torch.zeros(4, 96), torch.zeros(96, 96))


def test_descriptor_layouts_match_the_header():
    hdr = open(hip.HEADER_PATH if hasattr(hip, "HEADER_PATH") else "include/sdformerflow_hip.h").read()
    for name, cls, fields in (("SdfLinearDwDesc", hip.LinearDwDesc, ["dy", "x", "dw", "partial", "M", "N", "K", "nsplit", "cv_C", "cv_Wp"]),
                              ("SdfLinearTrainDesc", hip.LinearTrainDesc, ["a", "w", "bias", "out", "M", "N", "K", "mode", "cv_C", "cv_Wp"])):
        body = hdr[hdr.index("typedef struct " + name):hdr.index("} " + name + ";")]
        pos = [body.index(f) for f in ("* " + fields[0], *(f" {f}" for f in fields[1:]))]
        assert pos == sorted(pos), name                                       # same member order as the C declaration
        assert [f[0] for f in cls._fields_] == fields
    assert C.sizeof(hip.LinearDwDesc) == 4 * 8 + 8 + 5 * 4 + 4 and C.sizeof(hip.LinearTrainDesc) == 4 * 8 + 6 * 4
