"""Forward and dX of the training path's Linear layers on csrc/linear_train.hip (`sdf_linear_train_fwd`): products of the fp32 tensors
autograd holds, operands split into exact bf16 planes inside the kernel.  Checked against the same products in fp64; the bound is
stated against sum |a| |w| per element (what fp32 accumulation of exact products can lose), gradients spanning twelve decades."""
import pytest
import torch

from sdformerflow_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("M,N,K", [(1000, 96, 96), (130, 384, 96), (4097, 96, 384), (257, 192, 768), (64, 768, 3072), (5, 96, 96)])
def test_forward_on_spikes(M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + N)
    x = (torch.rand((M, K), generator=g) < 0.3).float().to(DEV)
    w = (torch.randn((N, K), generator=g) * 0.05).to(DEV)
    b = (torch.randn((N,), generator=g) * 0.1).to(DEV)
    out = hip.linear_train(x, w, b, mode=0)
    ref = x.double() @ w.double().t() + b.double()
    bound = x.double() @ w.double().abs().t() + b.double().abs()
    err = (out.double() - ref).abs()
    assert bool((err <= 1e-6 * bound + 1e-30).all()), float((err / (bound + 1e-30)).max())
    assert torch.equal(hip.linear_train(x, w, None, mode=0) + b, out) or float((hip.linear_train(x, w, None, mode=0) + b - out).abs().max()) < 1e-6


@pytest.mark.parametrize("M,N,K", [(1000, 96, 96), (130, 384, 96), (4097, 96, 384), (257, 768, 192), (64, 3072, 768), (5, 96, 96)])
def test_dx_of_gradients_spanning_twelve_decades(M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M + K)
    dy = (torch.randn((M, N), generator=g) * torch.pow(10.0, -12 * torch.rand((M, N), generator=g))).to(DEV)
    w = (torch.randn((N, K), generator=g) * 0.05).to(DEV)
    dx = hip.linear_train(dy, w, mode=1)
    ref = dy.double() @ w.double()
    bound = dy.double().abs() @ w.double().abs()
    err = (dx.double() - ref).abs()
    assert bool((err <= 2e-6 * bound + 1e-37).all()), float((err / (bound + 1e-37)).max())


def test_shapes_outside_the_kernel_are_refused():
    x = torch.zeros((10, 80), device=DEV)
    w = torch.zeros((96, 80), device=DEV)
    with pytest.raises(hip.SdfError):
        hip.linear_train(x, w, mode=0)                                   # K % 32
    with pytest.raises(hip.SdfError):
        hip.linear_train(torch.zeros((10, 96), device=DEV), w, mode=1)   # K % 96


def test_linear_function_matches_autograd_of_f_linear():
    """`autograd.LinearHipFunction` (train._linear's default on the fp32 path): output, dX, dW, db against torch.autograd through
    F.linear in fp64 on the same inputs."""
    import torch.nn.functional as F
    from sdformerflow_amd.autograd import LinearHipFunction
    g = torch.Generator(device="cpu").manual_seed(31)
    x = (torch.rand((2, 40, 162, 192), generator=g) < 0.25).float().to(DEV)
    w = (torch.randn((96, 192), generator=g) * 0.05).to(DEV)
    b = (torch.randn((96,), generator=g) * 0.1).to(DEV)
    go = (torch.randn((2, 40, 162, 96), generator=g) * 1e-5).to(DEV)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    yd = F.linear(xd, wd, bd)
    yd.backward(go.double())
    x_, w_, b_ = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = LinearHipFunction.apply(x_, w_, b_)
    y.backward(go)
    for got, ref, tol in ((y, yd, 1e-6), (x_.grad, xd.grad, 2e-6), (w_.grad, wd.grad, 2e-6), (b_.grad, bd.grad, 2e-6)):
        assert got.shape == ref.shape
        assert float((got.double() - ref.detach()).abs().max()) <= tol * float(ref.detach().abs().max())


@pytest.mark.parametrize("imgs,Cin,Cout,H,W", [(3, 96, 96, 20, 37), (2, 192, 96, 9, 12), (1, 96, 192, 33, 5), (4, 96, 96, 70, 64)])
def test_conv_forward_on_ringed_rows(imgs, Cin, Cout, H, W):
    """The convolution form of the forward (ringed channels-last rows, the tap a row offset of the loader, back through unring_rows)
    against F.conv2d in fp64: every tap, the zero padding at the image edges, images side by side on the row grid, the bias."""
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(imgs * 7 + W)
    x = (torch.rand((imgs, Cin, H, W), generator=g) < 0.3).float()
    w = torch.randn((Cout, Cin, 3, 3), generator=g) * 0.05
    b = torch.randn((Cout,), generator=g) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    bound = F.conv2d(x.double(), w.double().abs(), b.double().abs(), 1, 1)
    xr = hip._ringed_rows(x.to(DEV))
    y = hip.conv3x3_fwd_ringed(xr, w.to(DEV), b.to(DEV), imgs, H, W).cpu().double()
    assert y.shape == ref.shape
    err = (y - ref).abs()
    assert bool((err <= 1e-6 * bound + 1e-30).all()), float((err / (bound + 1e-30)).max())
    y0 = hip.conv3x3_fwd_ringed(xr, w.to(DEV), None, imgs, H, W).cpu().double()
    assert float((y0 + b.double().view(1, -1, 1, 1) - y).abs().max()) < 1e-6


def test_conv_function_matches_autograd_of_conv2d():
    import torch.nn.functional as F
    from sdformerflow_amd.autograd import Conv3x3HipFunction
    g = torch.Generator(device="cpu").manual_seed(77)
    x = (torch.rand((4, 96, 24, 32), generator=g) < 0.2).float()
    w = torch.randn((96, 96, 3, 3), generator=g) * 0.03
    go = torch.randn((4, 96, 24, 32), generator=g) * 1e-3
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    F.conv2d(xd, wd, None, 1, 1).backward(go.double())
    x_, w_ = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
    y = Conv3x3HipFunction.apply(x_, w_, None)
    y.backward(go.to(DEV))
    yd = F.conv2d(x.double(), w.double(), None, 1, 1)
    assert float((y.detach().cpu().double() - yd).abs().max()) <= 1e-6 * float(yd.abs().max())
    assert float((x_.grad.cpu().double() - xd.grad).abs().max()) <= 1e-4 * float(xd.grad.abs().max())      # MIOpen's Winograd dX
    assert float((w_.grad.cpu().double() - wd.grad).abs().max()) <= 2e-6 * float(wd.grad.abs().max())
