"""Weight gradient of a spike-fed Linear layer (csrc/linear_dw.hip, `sdf_linear_dw_fwd`): dW = dY^T X, what autograd computes for
nn.Linear in the reference's training step (train_flow_parallel_supervised_SNN.py:233-336).  Checked against the same product in
fp64 on the SAME inputs: the kernel's three-plane bf16 split of dY is exact, so what remains is fp32 accumulation - the bound is
stated against sum_m |dY| |X| per element, and gradient-like inputs span twelve decades."""
import ctypes

import pytest
import torch

from sdformerflow_amd import hip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(M, N, K, seed, decades=12.0, rate=0.3):
    g = torch.Generator(device="cpu").manual_seed(seed)
    dy = torch.randn((M, N), generator=g) * torch.pow(10.0, -decades * torch.rand((M, N), generator=g))
    x = (torch.rand((M, K), generator=g) < rate).float()
    return dy.to(DEV), x.to(DEV)


@pytest.mark.parametrize("M,N,K", [(1000, 96, 96), (33, 96, 192), (4096, 192, 96), (20000, 96, 384), (2500, 384, 96), (7, 96, 96),
                                   (4320, 768, 768)])
def test_dw_equals_the_fp64_product(M, N, K):
    dy, x = _inputs(M, N, K, 1000 + M)
    dw = hip.linear_dw(dy, x)
    ref = dy.double().t() @ x.double()
    bound = dy.double().abs().t() @ x.double()
    err = (dw.double() - ref).abs()
    # fp32 accumulation of exact products: a few ulp of the running sum per chunk; 2e-6 of the sum of magnitudes is generous
    assert bool((err <= 2e-6 * bound + 1e-37).all()), float((err / (bound + 1e-37)).max())
    # and, as the training tests state it, against the largest element
    assert float(err.max()) <= 2e-6 * float(ref.abs().max())


def test_the_split_of_dy_is_exact():
    """One row, one spike per column: dW[n, k] must be dY[0, n] bit for bit wherever x[0, k] = 1 - hi + mid + lo == dY."""
    N = K = 96
    g = torch.Generator(device="cpu").manual_seed(5)
    dy = (torch.randn((1, N), generator=g) * torch.pow(10.0, -30 * torch.rand((1, N), generator=g))).to(DEV)
    x = torch.ones((1, K), device=DEV)
    dw = hip.linear_dw(dy, x)
    assert torch.equal(dw, dy.t().expand(N, K))


def test_m_ranges_are_summed_in_a_fixed_order():
    dy, x = _inputs(50000, 96, 96, 77)
    assert hip.lib().sdf_linear_dw_splits(ctypes.c_int64(50000), 96, 96, 0) > 1
    a, b = hip.linear_dw(dy, x), hip.linear_dw(dy, x)
    assert torch.equal(a, b)


def test_shapes_outside_the_kernel_are_refused():
    dy, x = _inputs(64, 96, 80, 3)
    with pytest.raises(hip.SdfError):
        hip.linear_dw(dy, x)
    with pytest.raises(hip.SdfError):
        hip.linear_dw(dy.cpu(), x.cpu())


def test_linear_function_gradients_equal_autograd_of_f_linear():
    """`autograd.LinearDwFunction` (the training path's Linear on spikes, train._linear) against torch.autograd through
    F.linear on the same inputs: same output bits, same dX bits (both library products), dW / db within fp32 accumulation."""
    import torch.nn.functional as F
    from sdformerflow_amd.autograd import LinearDwFunction
    g = torch.Generator(device="cpu").manual_seed(9)
    x = (torch.rand((2, 50, 162, 96), generator=g) < 0.25).float().to(DEV)
    w = (torch.randn((384, 96), generator=g) * 0.05).to(DEV)
    b = (torch.randn((384,), generator=g) * 0.1).to(DEV)
    go = (torch.randn((2, 50, 162, 384), generator=g) * 1e-4).to(DEV)
    res = []
    for fn in (lambda x_, w_, b_: F.linear(x_, w_, b_), LinearDwFunction.apply):
        x_, w_, b_ = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = fn(x_, w_, b_)
        y.backward(go)
        res.append((y.detach(), x_.grad, w_.grad, b_.grad))
    (y0, gx0, gw0, gb0), (y1, gx1, gw1, gb1) = res
    assert torch.equal(y0, y1) and torch.equal(gx0, gx1)
    ref = go.reshape(-1, 384).double().t() @ x.reshape(-1, 96).double()
    assert float((gw1.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    assert float((gw0.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())      # (the library's own distance, for scale)
    assert torch.allclose(gb0, gb1, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("imgs,Cin,Cout,H,W", [(3, 96, 96, 20, 37), (2, 192, 96, 9, 12), (1, 96, 192, 33, 5), (5, 96, 96, 64, 48)])
def test_conv_weight_gradient_equals_autograd_in_fp64(imgs, Cin, Cout, H, W):
    """The convolution form (zero-ringed channels-last rows, taps as row offsets) against torch.autograd's conv2d weight gradient
    in fp64 on the CPU - every tap, the image edges, and the wrap of a row offset from one image into the next."""
    import torch.nn.functional as F
    g = torch.Generator(device="cpu").manual_seed(imgs * 100 + H)
    x = (torch.rand((imgs, Cin, H, W), generator=g) < 0.3).float()
    dy = torch.randn((imgs, Cout, H, W), generator=g) * torch.pow(10.0, -8 * torch.rand((imgs, Cout, H, W), generator=g))
    w = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w, None, 1, 1).backward(dy.double())
    ref = w.grad
    dw = hip.conv3x3_dw(dy.to(DEV), x.to(DEV)).cpu().double()
    assert dw.shape == ref.shape
    bound = torch.zeros_like(ref)
    wb = torch.zeros((Cout, Cin, 3, 3), dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), wb, None, 1, 1).backward(dy.double().abs())
    err = (dw - ref).abs()
    assert bool((err <= 2e-6 * wb.grad + 1e-37).all()), float((err / (wb.grad + 1e-37)).max())
    assert float(ref.abs().max()) > 0


def test_conv_function_gradients_equal_autograd_of_conv2d():
    import torch.nn.functional as F
    from sdformerflow_amd.autograd import Conv3x3DwFunction
    g = torch.Generator(device="cpu").manual_seed(21)
    x = (torch.rand((4, 96, 24, 32), generator=g) < 0.2).float().to(DEV)
    w = (torch.randn((96, 96, 3, 3), generator=g) * 0.03).to(DEV)
    go = (torch.randn((4, 96, 24, 32), generator=g) * 1e-3).to(DEV)
    res = []
    for fn in (lambda x_, w_: F.conv2d(x_, w_, None, 1, 1), lambda x_, w_: Conv3x3DwFunction.apply(x_, w_, None)):
        x_, w_ = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = fn(x_, w_)
        y.backward(go)
        res.append((y.detach(), x_.grad, w_.grad))
    (y0, gx0, gw0), (y1, gx1, gw1) = res
    assert torch.equal(y0, y1)
    assert torch.allclose(gx0, gx1, rtol=1e-4, atol=1e-7)              # (two library calls; the algorithm may differ between them)
    assert float((gw0 - gw1).abs().max()) <= 1e-4 * float(gw0.abs().max())
