"""Linear layer on two fp16 planes per operand (csrc/dense_linear.hip) against torch's fp64 matmul: the qkv / proj / fc1 / fc2
projections of the ANN swin block (reference models/STSwinNet/swin_transformer3D_v2.py:15-34, 176-202, 272-313) with bias, GELU
and the shortcut add in the epilogue."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [
    # M, K, N, bias, gelu, resid
    (1000, 96, 288, True, False, False),        # qkv, ragged last row tile
    (4096, 96, 96, True, False, True),          # proj + shortcut
    (777, 96, 384, True, True, False),          # fc1 + GELU
    (2048, 384, 96, True, False, True),         # fc2 + shortcut
    (640, 1536, 384, False, False, False),      # stage-2 fc2 shape, no bias
    (128, 32, 96, True, True, True),            # one tile, one k chunk
]


@pytest.mark.parametrize("M,K,N,bias,gelu,res", CASES)
def test_dense_linear_matches_fp64(M, K, N, bias, gelu, res):
    from sdformerflow_amd import hip
    g = torch.Generator().manual_seed(M + K + N)
    a = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = torch.randn(N, generator=g).cuda() if bias else None
    r = torch.randn(M, N, generator=g).cuda() if res else None
    assert hip.dense_linear_applicable(M, N, K)
    y = hip.dense_linear(a, hip.pack_dense_linear_weight(w), b, gelu, r)
    ref = a.double().cpu() @ w.double().cpu().T
    if bias:
        ref = ref + b.double().cpu()
    if gelu:
        ref = F.gelu(ref)
    if res:
        ref = ref + r.double().cpu()
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 4e-6 * ref.abs().max().item(), (err, ref.abs().max().item())


def test_in_place_shortcut_and_rejections():
    from sdformerflow_amd import hip
    g = torch.Generator().manual_seed(3)
    a = torch.randn(300, 96, generator=g).cuda()
    w = (torch.randn(96, 96, generator=g) / 10).cuda()
    x = torch.randn(300, 96, generator=g).cuda()
    ref = x + F.linear(a, w)
    out = hip.dense_linear(a, hip.pack_dense_linear_weight(w), None, False, x, out=x)       # x += a @ w.T
    assert out.data_ptr() == x.data_ptr()
    assert (x - ref).abs().max().item() <= 1e-5
    with pytest.raises(hip.SdfError):
        hip.dense_linear(a, hip.pack_dense_linear_weight(torch.randn(64, 96).cuda()))          # N % 96
    with pytest.raises(hip.SdfError):
        hip.dense_linear(torch.randn(10, 48).cuda(), hip.pack_dense_linear_weight(torch.randn(96, 48).cuda()))   # K % 32


@pytest.mark.parametrize("rows,C", [(1000, 96), (333, 192), (100, 384), (64, 768), (17, 1536), (5, 100), (9, 2048)])
def test_layer_norm_matches_torch(rows, C):
    """sdf_layer_norm_fwd against nn.LayerNorm in fp64 (reference swin_transformer3D_v2.py: norm1 / norm2 / PatchMerging.norm / norm{i})."""
    from sdformerflow_amd import hip
    g = torch.Generator().manual_seed(rows + C)
    x = (torch.randn(rows, C, generator=g) * 3 + 1.5).cuda()
    w, b = (0.5 + torch.rand(C, generator=g)).cuda(), torch.randn(C, generator=g).cuda()
    y = hip.layer_norm(x, w, b, 1e-5)
    ref = F.layer_norm(x.double().cpu(), (C,), w.double().cpu(), b.double().cpu(), 1e-5)
    assert (y.double().cpu() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()


@pytest.mark.parametrize("imgs,H,W,C,N,stride,T", [(4, 24, 32, 96, 96, 4, 2), (2, 17, 21, 32, 192, 2, 0), (6, 12, 12, 64, 96, 1, 3)])
def test_strided_convolution_as_gathered_gemm(imgs, H, W, C, N, stride, T):
    """The convolution form of sdf_dense_linear_fwd (PatchEmbedLocal.proj, reference models/STSwinNet/PatchEmbed.py:191) against
    F.conv2d in fp64, including the (T, B) -> (B, T) image order of the output."""
    from sdformerflow_amd import hip
    g = torch.Generator().manual_seed(imgs * H + C)
    x = torch.randn(imgs, C, H, W, generator=g).cuda()
    w = (torch.randn(N, C, 3, 3, generator=g) / (3 * C ** 0.5)).cuda()
    b = torch.randn(N, generator=g).cuda()
    wp = hip.pack_dense_linear_weight(w.permute(0, 2, 3, 1).reshape(N, -1))
    y = hip.dense_conv3x3_strided(x.permute(0, 2, 3, 1).contiguous(), wp, b, stride, T)
    ref = F.conv2d(x.double().cpu(), w.double().cpu(), b.double().cpu(), stride, 1).permute(0, 2, 3, 1)
    if T:
        ref = ref.reshape(T, imgs // T, *ref.shape[1:]).transpose(0, 1).reshape(ref.shape)
    assert y.shape == ref.shape
    err = (y.double().cpu() - ref).abs().max().item()
    assert err <= 4e-6 * ref.abs().max().item(), (err, ref.abs().max().item())
