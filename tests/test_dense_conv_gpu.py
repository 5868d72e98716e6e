"""Dense 3x3 convolution on two fp16 planes per operand (csrc/dense_conv_wres.hip) against torch's fp64 convolution on the CPU:
the convolutions of reference models/STSwinNet/PatchEmbed.py:166-196 / models/submodules.py:160-229 with BatchNorm folded,
residual add and ReLU in the epilogue."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _hip():
    from sdformerflow_amd import hip
    return hip


def _ref(x, w, alpha, beta, resid, relu):
    y = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
    if alpha is not None:
        y = y * alpha.double().cpu().view(1, -1, 1, 1)
    if beta is not None:
        y = y + beta.double().cpu().view(1, -1, 1, 1)
    if resid is not None:
        y = y + resid.double().cpu()
    return torch.relu(y) if relu else y


@pytest.mark.parametrize("scale", [1.0, 1e-3, 300.0])
def test_planes_round_trip(scale):
    hip = _hip()
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(3, 21, 9, 13, generator=g) * scale).cuda()
    p = hip.pack_planes(x)
    assert p.shape == (3, 2, 9, 13, 32)
    y = hip.unpack_planes(p, 21)
    # hi + lo carries 22 significant bits; below the fp16 normal range of lo the error is the subnormal spacing 2^-25
    assert torch.all((y - x).abs() <= x.abs() * 2.0 ** -21 + 2.0 ** -24)
    # padding channels of the last record are zero
    assert torch.count_nonzero(p.view(3, 2, 9, 13, 4, 2, 4)[:, 1, :, :, 2:, :, :].float()) == 0
    assert torch.count_nonzero(p.view(3, 2, 9, 13, 4, 2, 4)[:, 1, :, :, 1, :, 1:].float()) == 0


CASES = [
    # imgs, Cin, Cout, H, W, alpha/beta, resid, relu, out_f32
    (2, 96, 96, 24, 48, True, True, True, False),
    (2, 96, 96, 24, 48, True, False, True, False),
    (1, 96, 96, 13, 21, False, False, False, False),      # ragged tiles
    (3, 96, 96, 17, 35, True, True, True, True),          # fp32 channels-last output
    (2, 10, 96, 24, 40, False, False, False, False),      # head: one 16-channel record, 10 real channels
    (2, 16, 64, 9, 16, True, False, True, True),
    (1, 96, 32, 8, 16, True, True, False, False),         # one column block, one tile
]


@pytest.mark.parametrize("imgs,Cin,Cout,H,W,affine,res,relu,f32", CASES)
def test_dense_conv_matches_fp64(imgs, Cin, Cout, H, W, affine, res, relu, f32):
    hip = _hip()
    g = torch.Generator().manual_seed(imgs * 1000 + Cin + H)
    x = torch.randn(imgs, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    alpha = (0.5 + torch.rand(Cout, generator=g)).cuda() if affine else None
    beta = torch.randn(Cout, generator=g).cuda() if affine else None
    resid = torch.randn(imgs, Cout, H, W, generator=g).cuda() if res else None
    assert hip.dense_conv_applicable(imgs, H, W, Cin, Cout)
    out = hip.dense_conv3x3(hip.pack_planes(x), hip.pack_dense_conv_weight(w), alpha, beta,
                            hip.pack_planes(resid) if res else None, relu, f32)
    y = out.permute(0, 3, 1, 2) if f32 else hip.unpack_planes(out, Cout)
    ref = _ref(x, w, alpha, beta, resid, relu)
    err = (y.double().cpu() - ref).abs().max().item()
    # three fp16 products keep ~21 bits of each factor; the stored result is fp32 or a 22-bit hi + lo pair
    assert err <= 4e-6 * ref.abs().max().item(), (err, ref.abs().max().item())


def test_small_and_large_magnitudes():
    """Values near the fp16 ceiling keep the relative accuracy; small ones meet the absolute floor of the format: a lo half below
    the fp16 normal range (|a| < 0.12) is a subnormal with spacing 2^-24, which the matrix cores multiply without flushing."""
    hip = _hip()
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(32, 96, 3, 3, generator=g) / 30).cuda()
    wp = hip.pack_dense_conv_weight(w)
    for scale in (1e-3, 1.0, 2e3):
        x = (torch.randn(1, 96, 16, 16, generator=g) * scale).cuda()
        y = hip.unpack_planes(hip.dense_conv3x3(hip.pack_planes(x), wp), 32)
        ref = _ref(x, w, None, None, None, False)
        err = (y.double().cpu() - ref).abs().max().item()
        assert err <= 4e-6 * ref.abs().max().item() + 2e-7, (scale, err, ref.abs().max().item())


def test_rejects_unsupported_shapes():
    hip = _hip()
    x = hip.pack_planes(torch.randn(1, 32, 8, 16).cuda())               # two records: no instantiation
    w = hip.pack_dense_conv_weight(torch.randn(32, 32, 3, 3).cuda())
    with pytest.raises(hip.SdfError):
        hip.dense_conv3x3(x, w)
    x = hip.pack_planes(torch.randn(1, 16, 8, 16).cuda())
    w = hip.pack_dense_conv_weight(torch.randn(48, 16, 3, 3).cuda())     # 48 output channels: not a multiple of 32
    with pytest.raises(hip.SdfError):
        hip.dense_conv3x3(x, w)


@pytest.mark.parametrize("channels_last", [False, True])
def test_pack_planes_up2_is_bilinear_interpolation(channels_last):
    """sdf_pack_planes_up2 = F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) written as planes, for any input
    strides, into a record range of a wider tensor (reference models/submodules.py:117-157)."""
    hip = _hip()
    g = torch.Generator().manual_seed(5)
    a = torch.randn(2, 32, 9, 13, generator=g).cuda()
    b = torch.randn(2, 2, 9, 13, generator=g).cuda()
    if channels_last:
        a = a.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    planes = torch.zeros((2, 3, 18, 26, 32), dtype=torch.float16, device="cuda")
    hip.pack_planes_up2(a, planes, 0)
    hip.pack_planes_up2(b, planes, 2)
    got = hip.unpack_planes(planes, 48)
    ref = F.interpolate(torch.cat([a, b], 1), scale_factor=2, mode="bilinear", align_corners=False)
    # the planes keep 22 bits of each value; the interpolation itself rounds like torch's up to the order of its four products
    assert torch.all((got[:, :34] - ref).abs() <= ref.abs() * 2.0 ** -21 + 1e-6)
    assert torch.count_nonzero(got[:, 34:]) == 0


def test_pack_planes_zero_up2_is_the_packed_zero_upsampled_image():
    """sdf_pack_planes_zero_up2 (round 6: the SEW decoders' transposed convolutions as stride-1 correlations): the planes it writes are,
    bit for bit, sdf_pack_planes of the zero-filled image with x on the even pixels - channels-last parts (16-byte pieces), a strided
    2-of-32 channel slice (the prediction) and a plain NCHW tensor, each into its record range of a wider tensor."""
    hip = _hip()
    g = torch.Generator().manual_seed(6)
    a = torch.randn(3, 9, 13, 32, generator=g).cuda().permute(0, 3, 1, 2)               # channels-last view, C % 4 == 0
    p32 = torch.randn(3, 9, 13, 32, generator=g).cuda()
    b = p32[..., :2].permute(0, 3, 1, 2)                                                # 2 of 32 channels, channel stride 1, pixel stride 32
    c = torch.randn(3, 20, 9, 13, generator=g).cuda()                                   # NCHW
    planes = torch.full((3, 6, 18, 26, 32), 7.0, dtype=torch.float16, device="cuda")
    hip.pack_planes_zero_up2(a, planes, 0)                                              # records 0 - 1
    hip.pack_planes_zero_up2(b, planes, 2)                                              # record 2
    hip.pack_planes_zero_up2(c, planes, 3)                                              # records 3 - 4
    planes[:, 5:].zero_()
    up = torch.zeros((3, 96, 18, 26), device="cuda")
    up[:, 0:32, ::2, ::2] = a
    up[:, 32:34, ::2, ::2] = b
    up[:, 48:68, ::2, ::2] = c
    assert torch.equal(planes, hip.pack_planes(up))


@pytest.mark.parametrize("Cin,Cout", [(194, 96), (386, 96), (768, 192)])
def test_wide_convolution_as_a_chain_of_slices(Cin, Cout):
    """The decoders' convolutions (194, 386, 768 input channels) as chains of 96-channel slices plus one 16-channel record."""
    hip = _hip()
    g = torch.Generator().manual_seed(Cin)
    x = torch.randn(2, Cin, 12, 20, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3.0 * Cin ** 0.5)).cuda()
    bias = torch.randn(Cout, generator=g).cuda()
    xp = hip.pack_planes(x)
    slices = hip.dense_conv_slices(xp.shape[1])
    assert slices is not None and sum(n for _, n in slices) == xp.shape[1]
    wpad = F.pad(w, (0, 0, 0, 0, 0, xp.shape[1] * 16 - Cin))
    wsl = [((r0, n), hip.pack_dense_conv_weight(wpad[:, 16 * r0:16 * (r0 + n)])) for r0, n in slices]
    y = hip.dense_conv3x3_wide(xp, wsl, bias, True, True).permute(0, 3, 1, 2)
    ref = torch.relu(F.conv2d(x.double().cpu(), w.double().cpu(), bias.double().cpu(), 1, 1))
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 4e-6 * ref.abs().max().item(), (err, ref.abs().max().item())
    assert hip.dense_conv_slices(8) is None                            # 6 + 2: no chain of sixes and one single
