"""GPU parity tests: each HIP kernel, through the C ABI, against the CPU oracle on identical inputs.
Spikes are compared bit-for-bit; fp32 GEMM outputs to 1e-5 relative (tolerance stated per test)."""
import os

import numpy as np
import pytest
import torch

from oracle import neuron_ref as R
from oracle import sdformer_oracle as O
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def neuron_input(T, N=4096, seed=0):
    x = rnd((T, N), 100 + T + seed, -0.3, 0.6)
    x[:, :64] = 0.1
    x[0, 64:128] = 0.2
    return x


@pytest.mark.parametrize("T", [1, 2, 4, 7, 10, 20])
@pytest.mark.parametrize("v_reset", [None, 0.0, 0.05])
def test_lif_bit_exact(T, v_reset):
    x = neuron_input(T)
    ref, vref = R.neuron_ref(x, "lif", 2.0, 0.1, v_reset, return_aux=True)
    for dt in (torch.float32, torch.uint8):
        s, v = hip.lif_fwd(x.to(DEV), 2.0, 0.1, v_reset, dt, return_v=True)
        assert torch.equal(s.cpu().float(), ref)
        assert torch.equal(v.cpu(), vref)


@pytest.mark.parametrize("N", [1, 3, 243, 2 * 5 * 81 * 3 + 1])
def test_lif_and_psn_any_per_step_size_through_the_c_abi(N):
    """sdf_lif_fwd / sdf_psn_fwd with N % 4 != 0 (e.g. the (T', B_, 81, nH) token gate): served by the library itself - a C
    caller needs no padding; bit-equal to the C oracle, spikes and final membrane."""
    for T in (2, 7, 10):
        x = rnd((T, N), 300 + N + T, -0.3, 0.6)
        for v_reset in (None, 0.05):
            ref, vref = R.neuron_ref(x, "lif", 2.0, 0.1, v_reset, return_aux=True)
            s, v = hip.lif_fwd(x.to(DEV), 2.0, 0.1, v_reset, torch.uint8, return_v=True)
            assert torch.equal(s.cpu().float(), ref) and torch.equal(v.cpu(), vref)
        W, b = rnd((T, T), 5 + T, -0.5, 0.5) + 0.5 * torch.eye(T), torch.full((T, 1), -0.1)
        assert torch.equal(hip.psn_fwd(x.to(DEV), W.to(DEV), b.to(DEV)).cpu(), R.neuron_ref(x, "psn", psn_w=W, psn_b=b))


def test_lif_non_power_of_two_tau():
    x = neuron_input(10)
    ref = R.neuron_ref(x, "lif", 3.0, 0.1, None)
    assert torch.equal(hip.lif_fwd(x.to(DEV), 3.0, 0.1, None).cpu(), ref)


@pytest.mark.parametrize("T", [2, 4, 10, 20])
def test_psn_bit_exact(T):
    x = neuron_input(T)
    W = rnd((T, T), 5 + T, -0.5, 0.5) + 0.5 * torch.eye(T)
    b = torch.full((T, 1), -0.1)
    ref = R.neuron_ref(x, "psn", psn_w=W, psn_b=b)
    s = hip.psn_fwd(x.to(DEV), W.to(DEV), b.to(DEV), torch.uint8)
    assert torch.equal(s.cpu().float(), ref)
    assert 0.05 < ref.mean() < 0.95


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_neuron_fused_bn_pe_channel_last(kind):
    """SN(BN(x)+PE) over T'=2 on a (T', rows, C) tensor: Spiking_swin_transformer3D.py:675-680."""
    Tq, B_, N1, Cc = 2, 6, 81, 96
    x = rnd((Tq, B_ * N1, Cc), 3, -1.0, 1.0)
    alpha, beta = rnd((Cc,), 4, 0.5, 1.5), rnd((Cc,), 5, -0.2, 0.2)
    pe = rnd((Tq, N1 * Cc), 6, -0.3, 0.3)
    W, b = rnd((Tq, Tq), 7, -0.5, 0.5) + 0.5 * torch.eye(Tq), torch.full((Tq,), -0.1)
    ref = R.neuron_ref(x, kind, 2.0, 0.1, None, psn_w=W, psn_b=b, alpha=alpha, beta=beta, inner=1, add=pe,
                       add_period=N1 * Cc)
    p = hip.NeuronParams(kind, 2.0, 0.1, None, W.to(DEV), b.to(DEV))
    out = torch.empty((Tq, B_ * N1, Cc), dtype=torch.uint8, device=DEV)
    n = B_ * N1 * Cc
    hip.neuron_fwd(x.to(DEV), out, Tq, 1, n, 0, n, 0, n, p, alpha=alpha.to(DEV), beta=beta.to(DEV), Cch=Cc, inner=1,
                   add=pe.to(DEV), add_st=N1 * Cc, add_period=N1 * Cc)
    assert torch.equal(out.cpu().float(), ref)


def test_neuron_fused_bn_nchw_and_strided_time():
    """BN over dim 2 of (T,B,C,H,W) then LIF; and the (B,D,...) layout where time is the second axis."""
    T, B, Cc, H, W = 10, 2, 8, 6, 10
    x = rnd((T, B, Cc, H, W), 8, -1.0, 1.0)
    alpha, beta = rnd((Cc,), 9, 0.5, 1.5), rnd((Cc,), 10, -0.2, 0.2)
    ref = R.neuron_ref(x, "lif", 2.0, 0.1, None, alpha=alpha, beta=beta, inner=H * W)
    p = hip.NeuronParams("lif", 2.0, 0.1, None)
    out = torch.empty_like(x, device=DEV)
    n = B * Cc * H * W
    hip.neuron_fwd(x.to(DEV), out, T, 1, n, 0, n, 0, n, p, alpha=alpha.to(DEV), beta=beta.to(DEV), Cch=Cc, inner=H * W)
    assert torch.equal(out.cpu(), ref)
    # (B, T, inner) memory, neuron over T (the MLP's x.permute(1,0,2,3,4), :845)
    xb = x.permute(1, 0, 2, 3, 4).contiguous()
    ni = Cc * H * W
    out2 = torch.empty(xb.shape, dtype=torch.uint8, device=DEV)
    hip.neuron_fwd(xb.to(DEV), out2, T, B, ni, T * ni, ni, T * ni, ni, p, alpha=alpha.to(DEV), beta=beta.to(DEV), Cch=Cc,
                   inner=H * W)
    assert torch.equal(out2.cpu().float().permute(1, 0, 2, 3, 4), ref)


@pytest.mark.parametrize("shape,shift", [((1, 4, 18, 21), (1, 4, 4)), ((2, 4, 9, 21), (0, 0, 0))])
def test_neuron_window_gather(shape, shift):
    """pad + roll + window_partition_v2 folded into the load (Spiking_swin_transformer3D.py:789-804, 670)."""
    B, D, H, W = shape
    Cc = 32
    x = rnd((B, D, H, W, Cc), 11, -0.5, 1.0)
    ws, ss = O.get_window_size((D, H, W), (2, 9, 9), shift)
    xs, B_ = O.gather_slices(x, ws, ss)                                    # (2*B_, 81, C)
    ref = O.lif_multistep(xs.view(2, B_ * 81, Cc), 2.0, 0.1, None)
    src, _ = O.slice_table(B, D, H, W, ws, ss)
    rowmap = torch.from_numpy(src.reshape(-1).astype(np.int32)).to(DEV)     # [t'*B_*81 + row]
    out = torch.empty((2, B_ * 81, Cc), dtype=torch.uint8, device=DEV)
    n = B_ * 81 * Cc
    hip.neuron_fwd(x.to(DEV), out, 2, 1, n, 0, 0, 0, n, hip.NeuronParams("lif", 2.0, 0.1, None), rowmap=rowmap, rowlen=Cc)
    assert torch.equal(out.cpu().float(), ref)


def test_split_weight_reconstructs_fp32():
    W = rnd((96, 384), 12, -0.2, 0.2)
    W[0, 0], W[0, 1] = 1.0, -3.0e-5
    planes = hip.split_weight(W.to(DEV), 3).cpu()
    rec = sum((planes[i].to(torch.int32) << 16).view(torch.float32).double() for i in range(3))
    assert (rec - W.double()).abs().max() <= 2.0 ** -24 * W.abs().max()


def test_split_weight_f16x2_carries_22_bits():
    W = rnd((96, 384), 12, -0.2, 0.2)
    W[0, 0], W[0, 1], W[0, 2] = 1.0, -3.0e-5, 1.0e-9
    planes = hip.split_weight(W.to(DEV), 2)
    inv = planes.sdf_acc_scale
    assert inv > 0 and np.log2(inv) == int(np.log2(inv))         # a power of two: the rescale is exact
    rec = planes.cpu().view(torch.float16).double().sum(0) * inv
    err = (rec - W.double()).abs()
    assert (err <= 2.0 ** -22 * W.double().abs() + 2.0 ** -24 * inv).all()    # 22 bits; fp16-subnormal floor for tiny weights
    assert float(planes.cpu().view(torch.float16).abs().max()) < 2.0 ** 15


PLANES = [2, 3]        # weight formats of the spike matmuls: 2 = fp16 hi+lo (scaled), 3 = bf16 hi+mid+lo


def spikes(shape, seed, rate=0.3):
    return (rnd(shape, seed, 0.0, 1.0) < rate).to(torch.uint8)


@pytest.mark.parametrize("ns", PLANES)
@pytest.mark.parametrize("M,N,K", [(162 * 4, 96, 96), (1000, 384, 96), (333, 96, 384), (256, 192, 768), (130, 64, 32)])
def test_spike_gemm_plain(M, N, K, ns):
    A = spikes((M, K), 20 + M)
    W = rnd((N, K), 21, -0.3, 0.3)
    ref = A.double() @ W.double().t()
    out = torch.full((M, N), float("nan"), device=DEV)
    hip.spike_gemm(A.to(DEV), hip.split_weight(W.to(DEV), ns), out, M, N, K)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item(), err          # fp32-grade: exact / 22-bit weights, fp32 accumulate


@pytest.mark.parametrize("ns", PLANES)
def test_spike_gemm_epilogue_bias_bn_resid_scatter(ns):
    M, N, K = 500, 96, 192
    A = spikes((M, K), 30)
    W, bias = rnd((N, K), 31, -0.3, 0.3), rnd((N,), 32, -0.1, 0.1)
    alpha, beta = rnd((N,), 33, 0.5, 1.5), rnd((N,), 34, -0.2, 0.2)
    perm = torch.randperm(M + 50, generator=torch.Generator().manual_seed(0))[:M].to(torch.int32)
    perm[::7] = -1                                              # dropped rows (window padding)
    resid = rnd((M + 50, N), 35)
    ref = resid.clone().double()
    y = (A.double() @ W.double().t() + bias.double()) * alpha.double() + beta.double()
    keep = perm >= 0
    ref[perm[keep].long()] = y[keep] + resid.double()[perm[keep].long()]
    out = resid.clone().to(DEV)
    hip.spike_gemm(A.to(DEV), hip.split_weight(W.to(DEV), ns), out, M, N, K, bias=bias.to(DEV), alpha=alpha.to(DEV),
                   beta=beta.to(DEV), resid=out, out_rowmap=perm.to(DEV))
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("ns", PLANES)
@pytest.mark.parametrize("B_,nH", [(4, 3), (2, 6), (1, 12)])
def test_spike_gemm_head_scramble(B_, nH, ns):
    """A addressed through Z[t,b,n,g*32+d] = E_flat[((((b*nH+g)*2+t)*81+n)*32+d] (:709-710)."""
    Tq, N1, hd = 2, 81, 32
    Cc = nH * hd
    E = spikes((Tq, B_, N1, Cc), 40 + nH)
    tab = torch.from_numpy(O.z_gather_table(B_, nH, Tq, N1, hd))
    Z = E.reshape(-1)[tab.reshape(-1)].view(Tq * B_ * N1, Cc)
    W = rnd((Cc, Cc), 41, -0.3, 0.3)
    ref = Z.double() @ W.double().t()
    out = torch.empty((Tq * B_ * N1, Cc), device=DEV)
    hip.spike_gemm(E.to(DEV), hip.split_weight(W.to(DEV), ns), out, Tq * B_ * N1, Cc, Cc, zg=(nH, Tq, B_, N1))
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_qk_gate(kind):
    Tq, rows, Cc, nH = 2, 81 * 5, 96, 3
    q, k = spikes((Tq, rows, Cc), 50, 0.2), spikes((Tq, rows, Cc), 51, 0.5)
    W, b = rnd((Tq, Tq), 52, 0.0, 0.2) + 0.1 * torch.eye(Tq), torch.full((Tq,), -0.5)
    a = q.float().view(Tq, rows, nH, 32).sum(-1)
    A = R.neuron_ref(a, kind, 2.0, 0.1, None, psn_w=W, psn_b=b)
    ref = k.float() * A.repeat_interleave(32, dim=-1)
    e = torch.empty((Tq, rows, Cc), dtype=torch.uint8, device=DEV)
    hip.qk_gate(q.to(DEV), k.to(DEV), e, Tq, rows, Cc, hip.NeuronParams(kind, 2.0, 0.1, None, W.to(DEV), b.to(DEV)))
    assert torch.equal(e.cpu().float(), ref)
    assert 0.02 < ref.mean() < 0.6


# ---------------------------------------------------------------- fused window attention (a9, a10)
@pytest.mark.parametrize("with_mask", [True, False])
def test_win_attn_ann_cosine_softmax(with_mask):
    """WindowAttention3D core vs the oracle (swin_transformer3D_v2.py:176-202); exact-fp32 MFMA.  Logits reach
    ~100 (logit_scale up to 100 x cosine), so fp32 rounding of the logit is amplified by exp to ~|logit|*2^-24 ~ 6e-6
    relative on both sides; tolerance 2e-5 absolute on O(1) outputs."""
    nH, N, nW, B = 3, 162, 4, 2
    Cc = nH * 32
    qkv = rnd((B * nW, N, 3 * Cc), 60, -1.0, 1.0)
    ls = torch.exp(torch.clamp(rnd((nH, 1, 1), 61, 1.5, 5.0), max=float(np.log(100.0))))
    bias = 16 * torch.sigmoid(rnd((nH, N, N), 62, -2.0, 2.0))
    mask = O.compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4)) if with_mask else None
    ref, _ = O.ann_attention_core(qkv, ls, bias, mask, nH)
    got = hip.win_attn_ann(qkv.to(DEV), ls.reshape(-1).contiguous().to(DEV), bias.to(DEV),
                           mask.to(DEV) if with_mask else None, nH)
    assert (got.cpu() - ref).abs().max().item() <= 2e-5


@pytest.mark.parametrize("with_mask", [True, False])
@pytest.mark.parametrize("nH", [3, 6])
def test_win_attn_sew_linear(with_mask, nH):
    """Spiking_BN_WindowAttention3D core vs the oracle (Spiking_swin_transformer3D.py:320-363): binary q,k,v in
    the raw head view, no softmax; integer-valued QK^T so only the bias add and the P.V sum round (1e-5 rel)."""
    Tq, N1, nW, B = 2, 81, 4, 1
    B_, Cc, N = B * nW, nH * 32, 162
    q, k, v = spikes((Tq, B_, N1, Cc), 70, 0.3), spikes((Tq, B_, N1, Cc), 71, 0.4), spikes((Tq, B_, N1, Cc), 72, 0.5)
    bias = rnd((nH, N, N), 73, -1.0, 1.0)
    mask = O.compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4)) if with_mask else None
    scale = 32 ** -0.5
    view = lambda t: t.float().reshape(B_, nH, N, 32)
    ref, _ = O.sew_attention_core(view(q), view(k), view(v), scale, bias, mask, Tq, N1)
    sc = torch.full((nH,), scale)
    got = hip.win_attn_sew(q.to(DEV), k.to(DEV), v.to(DEV), sc.to(DEV), bias.to(DEV), mask.to(DEV) if with_mask else None,
                           nH, Tq, B_, N1)
    assert (got.cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


# ---------------------------------------------------------------- spike GEMM with the fused neuron epilogue
@pytest.mark.parametrize("ns", PLANES)
@pytest.mark.parametrize("kind", ["lif", "psn"])
@pytest.mark.parametrize("T,B,HW,K,N", [(10, 2, 50, 96, 384), (10, 1, 37, 192, 96), (2, 1, 81 * 5, 96, 96), (4, 3, 20, 32, 64),
                                         (20, 1, 9, 96, 96), (5, 2, 33, 64, 32)])
def test_spike_gemm_fused_neuron(kind, T, B, HW, K, N, ns):
    """SN_T(BN(A W^T) [+ add]) as one kernel vs GEMM(fp64) -> C oracle neuron.  Rows are (b, t, hw) like the MLP
    hidden layer (Spiking_swin_transformer3D.py:170-174).  The pre-activation differs from the fp64 one by fp32
    rounding, so a spike may flip only when it is within that rounding of the threshold: rate bound 2e-4."""
    A = spikes((B, T, HW, K), 80 + T)
    W = rnd((N, K), 81, -0.3, 0.3)
    alpha, beta = rnd((N,), 82, 0.5, 1.5), rnd((N,), 83, -0.2, 0.2)
    add = rnd((T, 7, N), 84, -0.3, 0.3)
    Wn, bn = rnd((T, T), 85, -0.5, 0.5) + 0.5 * torch.eye(T), torch.full((T,), -0.1)
    h = ((A.double() @ W.double().t()) * alpha.double() + beta.double())               # (B,T,HW,N)
    pos = torch.arange(B * HW) % 7
    h = h + add.double()[:, pos].view(T, B, HW, N).permute(1, 0, 2, 3)
    x = h.permute(1, 0, 2, 3).float().contiguous()                                      # (T,B,HW,N)
    ref = R.neuron_ref(x, kind, 2.0, 0.1, None, psn_w=Wn, psn_b=bn).permute(1, 0, 2, 3)
    out = torch.zeros((B * T * HW, N), dtype=torch.uint8, device=DEV)
    p = hip.NeuronParams(kind, 2.0, 0.1, None, Wn.to(DEV), bn.to(DEV))
    hip.spike_gemm_sn(A.view(-1, K).to(DEV), hip.split_weight(W.to(DEV), ns), out, N, K, T, B * HW, HW, T * HW, HW, p,
                      alpha=alpha.to(DEV), beta=beta.to(DEV), add=add.to(DEV), add_prows=7)
    got = out.cpu().view(B, T, HW, N).float()
    rate = (got != ref).float().mean().item()
    assert rate <= 2e-4, rate
    assert 0.03 < ref.mean() < 0.97


# ---------------------------------------------------------------- spike convolution (implicit GEMM)
@pytest.mark.parametrize("ns", PLANES)
@pytest.mark.parametrize("imgs,H,W,Cin,Cout,stride", [(3, 20, 24, 48, 96, 2), (2, 18, 22, 96, 96, 1), (2, 9, 12, 192, 192, 1),
                                                      (10, 72, 96, 96, 96, 1)])
def test_spike_conv3x3_fp32_epilogue(imgs, H, W, Cin, Cout, stride, ns):
    """3x3 / pad 1 convolution on NHWC u8 spikes + BN + residual vs torch conv2d in fp64 (1e-5 relative)."""
    x = spikes((imgs, H, W, Cin), 90 + Cin)
    w = rnd((Cout, Cin, 3, 3), 91, -0.1, 0.1)
    alpha, beta = rnd((Cout,), 92, 0.5, 1.5), rnd((Cout,), 93, -0.2, 0.2)
    OH, OW = (H + 2 - 3) // stride + 1, (W + 2 - 3) // stride + 1
    resid = rnd((imgs * OH * OW, Cout), 94)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, stride, 1).permute(0, 2, 3, 1)
    ref = ref.reshape(-1, Cout) * alpha.double() + beta.double() + resid.double()
    out = torch.empty((imgs * OH * OW, Cout), device=DEV)
    hip.spike_conv2d(x.to(DEV), hip.pack_conv_weight(w.to(DEV), ns), imgs, H, W, Cin, OH, OW, 3, 3, stride, (-1, 0, 1), (-1, 0, 1),
                     out=out, alpha=alpha.to(DEV), beta=beta.to(DEV), resid=resid.to(DEV))
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("ns", PLANES)
@pytest.mark.parametrize("kind", ["lif", "psn"])
@pytest.mark.parametrize("B,H,W", [(2, 7, 9), (1, 72, 96)])
def test_spike_conv3x3_fused_neuron(kind, B, H, W, ns):
    """conv -> BN -> neuron over T=10 in one kernel (MS_ResBlock conv1 -> norm1 -> sn2, Spiking_modules.py:914-920).
    The 72 x 96 case runs several tiles per workgroup with both consumer groups and the producers busy - the
    configuration in which SLP-packed f32 FMAs once corrupted the PSN sums of the last 16 lanes (findings)."""
    T, Cc = 10, 96
    x = spikes((T * B, H, W, Cc), 95)
    w = rnd((Cc, Cc, 3, 3), 96, -0.1, 0.1)
    alpha, beta = rnd((Cc,), 97, 0.5, 1.5), rnd((Cc,), 98, -0.2, 0.2)
    Wn, bn = rnd((T, T), 85, -0.5, 0.5) + 0.5 * torch.eye(T), torch.full((T,), -0.1)
    h = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    h = (h * alpha.double() + beta.double()).float().view(T, B * H * W * Cc)
    ref = R.neuron_ref(h, kind, 2.0, 0.1, None, psn_w=Wn, psn_b=bn).view(T * B, H, W, Cc)
    n = B * H * W
    sn = hip.NeuronParams(kind, 2.0, 0.1, None, Wn.to(DEV), bn.to(DEV))
    outs = []
    for _ in range(2):
        out = torch.zeros((T * B * H * W, Cc), dtype=torch.uint8, device=DEV)
        hip.spike_conv2d(x.to(DEV), hip.pack_conv_weight(w.to(DEV), ns), T * B, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1),
                         out_spike=out, alpha=alpha.to(DEV), beta=beta.to(DEV), sn=sn, sn_T=T, pos=(n, n, 0, n))
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1])                         # deterministic run to run
    rate = (outs[0].view(T * B, H, W, Cc).float() != ref).float().mean().item()
    assert rate <= 2e-4, rate
    assert 0.03 < ref.mean() < 0.97


@pytest.mark.parametrize("ns", PLANES)
def test_spike_conv_transpose_parity_classes(ns):
    """ConvTranspose2d(k=3, s=2, p=1, output_padding=1) on spikes as four parity-class implicit GEMMs
    (MS_SpikingTransposeDecoderLayer, Spiking_modules.py:416-446) vs torch conv_transpose2d in fp64."""
    from sdformerflow_amd.engine import deconv_classes
    imgs, H, W, Cin, Cout = 3, 9, 12, 200, 96               # Cin padded to 208 like the decoder concat buffers
    cp = 208
    x = torch.zeros((imgs, H, W, cp), dtype=torch.uint8)
    x[..., :Cin] = spikes((imgs, H, W, Cin), 99)
    w = rnd((Cin, Cout, 3, 3), 100, -0.1, 0.1)
    ref = torch.nn.functional.conv_transpose2d(x[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), None, 2, 1, 1)
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    out = torch.full((imgs * 2 * H * 2 * W, Cout), float("nan"), device=DEV)
    for cls in deconv_classes(w.to(DEV), imgs, H, W, cp, ns, DEV):
        hip.spike_conv2d(x.to(DEV), cls["Wp"], imgs, H, W, cp, H, W, cls["KH"], cls["KW"], 1, cls["dy"], cls["dx"], out=out,
                         out_rowmap=cls["rowmap"])
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("ns", PLANES)
def test_spike_conv_small_m_split_k(ns):
    """Small-M / large-K convolution (U-Net res-block shape: 10 x 9 x 12 pixels, 768 -> 768 channels): the library
    splits K over workgroups and reduces the partial sums in k order; same 1e-5 bound, and bit-identical run to run."""
    imgs, H, W, Cc = 10, 9, 12, 768
    x = spikes((imgs, H, W, Cc), 110)
    w = rnd((Cc, Cc, 3, 3), 111, -0.05, 0.05)
    alpha, beta = rnd((Cc,), 112, 0.5, 1.5), rnd((Cc,), 113, -0.2, 0.2)
    resid = rnd((imgs * H * W, Cc), 114)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    ref = ref.reshape(-1, Cc) * alpha.double() + beta.double() + resid.double()
    Wp = hip.pack_conv_weight(w.to(DEV), ns)
    outs = []
    for _ in range(2):
        out = torch.empty((imgs * H * W, Cc), device=DEV)
        hip.spike_conv2d(x.to(DEV), Wp, imgs, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=out, alpha=alpha.to(DEV),
                         beta=beta.to(DEV), resid=resid.to(DEV))
        outs.append(out.cpu())
    assert (outs[0].double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert torch.equal(outs[0], outs[1])


# ---------------------------------------------------------------- the two ends of the forward
@pytest.mark.parametrize("kind", ["lif", "psn"])
@pytest.mark.parametrize("B,T,H,W,Cout", [(1, 10, 9, 32, 48), (2, 5, 6, 16, 32)])
def test_head_conv_bn_neuron(kind, B, T, H, W, Cout):
    """conv3x3(real-valued 2-channel voxel) -> BN -> neuron over T in one kernel vs conv2d(fp64) -> BN -> C oracle neuron
    (MS_PED_Spiking_PatchEmbed_Conv_sfn.head, Spiking_modules.py:1782).  fp32-vs-fp64 rounding may flip a spike only
    when the pre-activation is within rounding of the threshold."""
    x = rnd((B * T, H, W, 2), 120, -1.0, 3.0)
    x[x.abs() < 0.8] = 0.0                                               # sparse event counts
    w = rnd((Cout, 2, 3, 3), 121, -0.5, 0.5)
    alpha, beta = rnd((Cout,), 122, 0.5, 1.5), rnd((Cout,), 123, -0.2, 0.2)
    Wn, bn = rnd((T, T), 85, -0.5, 0.5) + 0.5 * torch.eye(T), torch.full((T,), -0.1)
    h = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    h = (h * alpha.double() + beta.double()).float().view(B, T, H * W * Cout).permute(1, 0, 2).contiguous()
    ref = R.neuron_ref(h.view(T, -1), kind, 2.0, 0.1, None, psn_w=Wn, psn_b=bn).view(T, B, H, W, Cout).permute(1, 0, 2, 3, 4)
    assert hip.head_conv_sn_supported(T, H, W, 2, Cout)
    p = hip.NeuronParams(kind, 2.0, 0.1, None, Wn.to(DEV), bn.to(DEV))
    out = hip.head_conv_sn(x.to(DEV), w.to(DEV), B, T, H, W, p, alpha=alpha.to(DEV), beta=beta.to(DEV))
    rate = (out.cpu().float() != ref).float().mean().item()
    assert rate <= 2e-4, rate
    assert 0.03 < ref.mean() < 0.97
    # the same input as the event voxel (B, bins, 2, H, W) read IN PLACE through strides (bins = T here, plus two spare bins the
    # kernel must not touch): bit-equal to the packed NHWC launch
    vox = torch.full((B, T + 2, 2, H, W), 7.0)
    vox[:, :T] = x.view(B, T, H, W, 2).permute(0, 1, 4, 2, 3)
    out2 = hip.head_conv_sn(vox.to(DEV), w.to(DEV), B, T, H, W, p, alpha=alpha.to(DEV), beta=beta.to(DEV), voxel_bins=T + 2)
    assert torch.equal(out2, out)


@pytest.mark.parametrize("h,w,H,W", [(9, 12, 288, 384), (36, 48, 288, 384), (5, 7, 20, 28)])
def test_flow_out_sum_and_nearest_upsample(h, w, H, W):
    B, D = 2, 10
    buf = rnd((B * D * h * w, 32), 130, -1.0, 1.0).to(DEV)
    pred = buf.view(B, D, h, w, 32)[..., :2]
    ref = torch.nn.functional.interpolate(pred.cpu().double().sum(1).permute(0, 3, 1, 2), scale_factor=(H / h, W / w))
    got = hip.flow_out(pred, H, W, H / h, W / w)
    assert got.shape == ref.shape
    assert (got.cpu().double() - ref).abs().max().item() <= 1e-5


@pytest.mark.parametrize("ns", PLANES)
def test_spike_conv_transpose_as_gemm_plus_col2im(ns):
    """The same ConvTranspose2d(k=3, s=2, p=1, output_padding=1) as ONE spike GEMM over the nine stacked tap matrices followed
    by the col2im + BN kernel (the form the engine uses for the small decoder levels) vs torch conv_transpose2d in fp64."""
    from sdformerflow_amd.engine import deconv_tap_weights
    imgs, H, W, Cin, Cout = 3, 9, 12, 200, 96
    cp = 224                                                     # Cin padded to a multiple of 32 (GEMM K)
    x = torch.zeros((imgs, H, W, cp), dtype=torch.uint8)
    x[..., :Cin] = spikes((imgs, H, W, Cin), 99)
    w = rnd((Cin, Cout, 3, 3), 100, -0.1, 0.1)
    alpha, beta = rnd((Cout,), 101, 0.5, 1.5), rnd((Cout,), 102, -0.2, 0.2)
    ref = torch.nn.functional.conv_transpose2d(x[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), None, 2, 1, 1)
    ref = ref.permute(0, 2, 3, 1) * alpha.double() + beta.double()
    Y = torch.empty((imgs * H * W, 9 * Cout), device=DEV)
    hip.spike_gemm(x.to(DEV), deconv_tap_weights(w.to(DEV), cp, ns), Y, imgs * H * W, 9 * Cout, cp)
    out = hip.deconv_col2im(Y, imgs, H, W, Cout, alpha=alpha.to(DEV), beta=beta.to(DEV))
    assert out.shape == (imgs, 2 * H, 2 * W, Cout)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()


@pytest.mark.parametrize("ns", PLANES)
@pytest.mark.parametrize("with_resid", [True, False])
def test_spike_conv3x3_fused_neuron_with_membrane_output(with_resid, ns):
    """conv -> BN -> (+ identity) stored as fp32 AND the T=10 neuron on that sum, one launch (MS_ResBlock conv2 + the next
    block's sn1, Spiking_modules.py:922-933): membrane within 1e-5 of fp64, spikes = C-oracle neuron of the kernel's own
    membrane bit for bit."""
    T, B, H, W, Cc = 10, 1, 36, 48, 96
    x = spikes((T * B, H, W, Cc), 140)
    w = rnd((Cc, Cc, 3, 3), 141, -0.1, 0.1)
    alpha, beta = rnd((Cc,), 142, 0.5, 1.5), rnd((Cc,), 143, -0.2, 0.2)
    resid = rnd((T * B * H * W, Cc), 144) if with_resid else None
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cc)
    ref = ref * alpha.double() + beta.double() + (resid.double() if with_resid else 0)
    n = B * H * W
    m = torch.full((T * n, Cc), float("nan"), device=DEV)
    sp = torch.zeros((T * n, Cc), dtype=torch.uint8, device=DEV)
    hip.spike_conv2d(x.to(DEV), hip.pack_conv_weight(w.to(DEV), ns), T * B, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=m,
                     out_spike=sp, alpha=alpha.to(DEV), beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV),
                     sn=hip.NeuronParams("lif", 2.0, 0.1, None), sn_T=T, pos=(n, n, 0, n))
    assert (m.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    want = R.neuron_ref(m.cpu().view(T, -1), "lif", 2.0, 0.1, None).view(T * n, Cc)
    assert torch.equal(sp.cpu().float(), want)


# ------------------------------------------------------------------ neuron backward (training path)
from oracle import neuron_bwd_ref as BW  # noqa: E402

NG = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "neuron_grads.npz"))


def _grad_inputs(T, N, seed):
    x = rnd((T, N), seed, -0.3, 0.6)
    x[:, :64] = 0.1
    x[0, 64:128] = 0.2
    return x, rnd((T, N), seed + 1000, -1.0, 2.0)


@pytest.mark.parametrize("T", [1, 2, 4, 5, 8, 10, 16, 20])
@pytest.mark.parametrize("v_reset,detach,tau", [(None, True, 2.0), (None, False, 2.0), (0.0, True, 2.0), (0.0, False, 2.0),
                                                 (0.05, False, 2.0), (None, True, 3.0)])
def test_lif_backward_matches_oracle(T, v_reset, detach, tau):
    """sdf_lif_bwd against the CPU restatement of the reference's autograd: bit-equal when the reset is detached and
    tau is a power of two (the shipped configuration), 1e-6 of the largest gradient otherwise (three-term sums)."""
    x, g = _grad_inputs(T, 1 << 14, 500 + T)
    ref = BW.lif_backward(x, g, tau, 0.1, v_reset, detach, 2.0)
    got = hip.lif_bwd(x.to(DEV), g.to(DEV), tau, 0.1, v_reset, detach, 2.0).cpu()
    if detach and tau == 2.0:
        assert torch.equal(got, ref)
    else:
        assert (got - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()


@pytest.mark.parametrize("T", [2, 10])
def test_lif_backward_matches_reference_fixture(T):
    from sdformerflow_amd.synthetic import synth_uniform
    x = synth_uniform((T, 2048), 300 + T, -0.3, 0.6)
    x[:, :64] = 0.1
    x[0, 64:128] = 0.2
    g = synth_uniform((T, 2048), 400 + T, -1.0, 2.0)
    for tag, v_reset, detach, tau in (("soft_detach", None, True, 2.0), ("soft_nodetach", None, False, 2.0),
                                      ("hard_detach", 0.0, True, 2.0), ("hard_nodetach", 0.0, False, 2.0),
                                      ("tau3_soft_detach", None, True, 3.0)):
        got = hip.lif_bwd(x.to(DEV), g.to(DEV), tau, 0.1, v_reset, detach, 2.0).cpu()
        ref = torch.from_numpy(NG[f"lif_{tag}_T{T}_gx"])
        if detach and tau == 2.0:
            assert torch.equal(got, ref), tag
        else:
            assert (got - ref).abs().max().item() <= 1e-6 * ref.abs().max().item(), tag


def test_if_backward_matches_oracle():
    x, g = _grad_inputs(10, 1 << 12, 77)
    ref = BW.lif_backward(x, g, 2.0, 0.1, None, True, 2.0, kind="if")
    got = hip.lif_bwd(x.to(DEV), g.to(DEV), 2.0, 0.1, None, True, 2.0, kind="if").cpu()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("T", [2, 5, 10, 20])
def test_psn_backward_matches_oracle(T):
    """sdf_psn_bwd: dx / dW / db against the CPU restatement (h through addmm there, through the fma chain here: elements
    whose h differs in the last bit change g'(h) by ~1e-7 relative); T = 20 takes the grad_h + library-GEMM route."""
    from sdformerflow_amd.synthetic import synth_state_dict
    N = 1 << 16
    x, g = _grad_inputs(T, N, 900 + T)
    sd = synth_state_dict({"spiking_neuron.weight": (T, T), "spiking_neuron.bias": (T, 1)}, salt=T)
    W, b = sd["spiking_neuron.weight"], sd["spiking_neuron.bias"]
    rx, rW, rb = BW.psn_backward(x, W, b, g, 2.0)
    gx, gW, gb = hip.psn_bwd(x.to(DEV), W.to(DEV), b.to(DEV), g.to(DEV), 2.0)
    assert (gx.cpu() - rx).abs().max().item() <= 1e-5 * rx.abs().max().item()
    assert (gW.cpu() - rW).abs().max().item() <= 1e-4 * rW.abs().max().item()
    assert (gb.cpu().view(-1) - rb.view(-1)).abs().max().item() <= 1e-4 * rb.abs().max().item()
    gx2, gW2, gb2 = hip.psn_bwd(x.to(DEV), W.to(DEV), b.to(DEV), g.to(DEV), 2.0)
    assert torch.equal(gW, gW2) and torch.equal(gb, gb2) and torch.equal(gx, gx2)       # deterministic reduction
    if T in (2, 10):
        small = hip.psn_bwd(*(t.to(DEV) for t in (_fixture_x(T), W, b, _fixture_g(T))), 2.0)
        for got, key in zip(small, ("gx", "gW", "gb")):
            ref = torch.from_numpy(NG[f"psn_T{T}_{key}"])
            assert (got.cpu().view(-1) - ref.view(-1)).abs().max().item() <= 1e-5 * ref.abs().max().item(), key


def _fixture_x(T):
    x = rnd((T, 2048), 300 + T, -0.3, 0.6)
    x[:, :64] = 0.1
    x[0, 64:128] = 0.2
    return x


def _fixture_g(T):
    return rnd((T, 2048), 400 + T, -1.0, 2.0)


def test_neuron_modules_are_differentiable_on_the_gpu():
    """The module-level surface the reference's training loop touches: `Spiking_neuron(...)(x)` under autograd gives the
    oracle's gradients (lif: bit-equal; psn incl. weight / bias grads) - forward and backward both on HIP kernels."""
    from sdformerflow_amd.STSwinNet_SNN.Spiking_modules import Spiking_neuron
    T = 10
    x0, g = _fixture_x(T), _fixture_g(T)
    n = Spiking_neuron(num_steps=T, neuron_type="lif", v_th=0.1, v_reset=None, surrogate_fun="surrogate.ATan()", tau=2.0,
                       detach_reset=True).to(DEV)
    x = x0.to(DEV).requires_grad_(True)
    s = n(x)
    s.backward(g.to(DEV))
    assert torch.equal(s.detach().cpu().to(torch.uint8), torch.from_numpy(NG[f"lif_soft_detach_T{T}_s"]))
    assert torch.equal(x.grad.cpu(), torch.from_numpy(NG[f"lif_soft_detach_T{T}_gx"]))
    from sdformerflow_amd.synthetic import synth_state_dict
    p = Spiking_neuron(num_steps=T, neuron_type="psn", v_th=0.1, surrogate_fun="surrogate.ATan()").to(DEV)
    sd = synth_state_dict({"spiking_neuron.weight": (T, T), "spiking_neuron.bias": (T, 1)}, salt=T)
    p.load_state_dict(sd)
    x = x0.to(DEV).requires_grad_(True)
    p(x).backward(g.to(DEV))
    for got, key in ((x.grad, "gx"), (p.spiking_neuron.weight.grad, "gW"), (p.spiking_neuron.bias.grad, "gb")):
        ref = torch.from_numpy(NG[f"psn_T{T}_{key}"])
        assert got.shape == ref.shape and (got.cpu() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item(), key


# ------------------------------------------------------------------ the attention as one C-ABI call (SURVEY.md 8b sdf_qk_attn_fwd)
@pytest.mark.parametrize("shape", [(1, 10, 72, 96, (2, 9, 9), (1, 4, 4)), (2, 4, 18, 21, (2, 9, 9), (0, 0, 0)),
                                   (1, 5, 16, 24, (2, 8, 8), (1, 4, 4)), (1, 20, 30, 40, (2, 15, 15), (1, 7, 7))])
def test_window_slice_map_built_on_the_device_equals_the_host_table(shape):
    """sdf_window_slice_map (index arithmetic in a kernel) against the numpy table that tests/test_oracle_golden.py pins on
    the reference's pad + roll + window_partition_v2 fixtures: bit-equal, padding rows -1."""
    from sdformerflow_amd.STSwinNet_SNN.Spiking_swin_transformer3D import window_slice_map
    B, D, H, W, ws, ss = shape
    ref, B_ref = window_slice_map(B, D, H, W, ws, ss)
    got, B_ = hip.window_slice_map(B, D, H, W, ws, ss, DEV)
    assert B_ == B_ref and np.array_equal(got.cpu().numpy(), ref.reshape(-1))


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_qk_attn_single_call_equals_its_four_launches(kind):
    """sdf_qk_attn_fwd (neuron -> q|k GEMM -> gate -> projection GEMM inside the library) is bit-equal to the same four
    entry points driven one by one from the host, for the stacked (lif) and the separate (psn) projection form; the block-level
    parity against the oracle is tests/test_engine_gpu.py (the engine goes through this call)."""
    import yaml
    from sdformerflow_amd.engine import _Block
    from sdformerflow_amd.STSwinNet_SNN import Spiking_swin_transformer3D as SW
    from sdformerflow_amd.synthetic import synth_state_dict
    kw = {"num_steps": 4, "v_reset": None, "v_th": 0.1, "neuron_type": kind, "surrogate_fun": "surrogate.ATan()", "tau": 2.0,
          "detach_reset": True, "spike_norm": "BN"}
    B, D, H, W, Cc, nH = 2, 4, 18, 21, 96, 3
    m = SW.MS_Spiking_SwinTransformerBlock3D(Cc, (H, W), nH, window_size=(2, 9, 9), shift_size=(1, 4, 4), norm_layer="BN", **kw)
    m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    blk = _Block(m.to(DEV).eval(), torch.device(DEV), 2)
    assert (blk.qk is not None) == (kind == "lif")
    x0 = rnd((B, D, H, W, Cc), 31, -0.5, 1.0).to(DEV)
    ws, ss = (2, 9, 9), (1, 4, 4)
    rowmap, B_ = hip.window_slice_map(B, D, H, W, ws, ss, DEV)
    Tq, N1 = 2, 81
    rows = B_ * N1
    M = Tq * rows
    # (a) four launches from the host
    xa = x0.clone()
    xs = torch.empty((M, Cc), dtype=torch.uint8, device=DEV)
    hip.neuron_fwd(xa, xs, Tq, 1, rows * Cc, 0, 0, 0, rows * Cc, blk.sn_proj, rowmap=rowmap, rowlen=Cc)
    if blk.qk is not None:
        qk = torch.empty((M, 2 * Cc), dtype=torch.uint8, device=DEV)
        hip.spike_gemm_sn(xs, blk.qk["Wp"], qk, 2 * Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_q, alpha=blk.qk["alpha"],
                          beta=blk.qk["beta"], add=blk.qk["add"], add_prows=N1)
        hip.qk_gate(qk, qk[:, Cc:], xs, Tq, rows, Cc, blk.sn2_q, ldq=2 * Cc, ldk=2 * Cc)
    else:
        q = torch.empty((M, Cc), dtype=torch.uint8, device=DEV)
        k = torch.empty((M, Cc), dtype=torch.uint8, device=DEV)
        hip.spike_gemm_sn(xs, blk.q.Wp, q, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_q, alpha=blk.q.alpha, beta=blk.q.beta)
        hip.spike_gemm_sn(xs, blk.k.Wp, k, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_k, alpha=blk.k.alpha, beta=blk.k.beta,
                          add=blk.pe, add_prows=N1)
        hip.qk_gate(q, k, xs, Tq, rows, Cc, blk.sn2_q)
    hip.spike_gemm(xs, blk.p.Wp, xa, M, Cc, Cc, bias=blk.p.bias, alpha=blk.p.alpha, beta=blk.p.beta, resid=xa, out_rowmap=rowmap,
                   zg=(nH, Tq, B_, N1))
    # (b) one call
    xb = x0.clone()
    if blk.qk is not None:
        hip.qk_attn(xb, rowmap, B_, Tq, N1, nH, blk.p, blk.sn_proj, blk.sn_q, blk.sn_k, blk.sn2_q, qk=blk.qk)
    else:
        hip.qk_attn(xb, rowmap, B_, Tq, N1, nH, blk.p, blk.sn_proj, blk.sn_q, blk.sn_k, blk.sn2_q, q_lin=blk.q, k_lin=blk.k, pe=blk.pe)
    torch.cuda.synchronize()
    assert torch.equal(xa, xb) and not torch.equal(xa, x0)


def test_spike_gemm_bn_plain_entry_point():
    import ctypes as C
    M, K, N = 4096, 96, 192
    A = (torch.rand((M, K), device=DEV) < 0.3).to(torch.uint8)
    Wf = torch.randn((N, K), device=DEV) * 0.1
    Wp = hip.split_weight(Wf, 2)
    a, b = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV)
    ref = torch.empty((M, N), device=DEV)
    hip.spike_gemm(A, Wp, ref, M, N, K, alpha=a, beta=b)
    out = torch.empty((M, N), device=DEV)
    rc = hip.lib().sdf_spike_gemm_bn_fwd(C.c_void_p(A.data_ptr()), C.c_void_p(Wp.data_ptr()), C.c_int(2), C.c_float(Wp.sdf_acc_scale),
                                         C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()),
                                         C.c_int64(M), C.c_int(K), C.c_int(N), hip._stream())
    torch.cuda.synchronize()
    assert rc == 0 and torch.equal(out, ref)


@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_ms_mlp_single_call_equals_its_three_launches(kind):
    """sdf_ms_mlp_fwd (three-launch form) against the same three entry points driven from the host (bit-equal); parity of the MLP against the
    oracle is the teacher-forced block test of tests/test_engine_gpu.py, which goes through this call."""
    from sdformerflow_amd.engine import _Block
    from sdformerflow_amd.STSwinNet_SNN import Spiking_swin_transformer3D as SW
    from sdformerflow_amd.synthetic import synth_state_dict
    B, D, H, W, Cc = 2, 10, 9, 12, 96
    kw = {"num_steps": D, "v_reset": None, "v_th": 0.1, "neuron_type": kind, "surrogate_fun": "surrogate.ATan()", "tau": 2.0,
          "detach_reset": True, "spike_norm": "BN"}
    m = SW.MS_Spiking_SwinTransformerBlock3D(Cc, (H, W), 3, window_size=(2, 9, 9), shift_size=(0, 0, 0), norm_layer="BN", **kw)
    m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    blk = _Block(m.to(DEV).eval(), torch.device(DEV), 2)
    x0 = rnd((B, D, H, W, Cc), 33, -0.5, 1.0).to(DEV)
    hw, ntok, Ch = H * W, B * D * H * W, 4 * Cc
    xa = x0.clone()
    s1 = torch.empty((ntok, Cc), dtype=torch.uint8, device=DEV)
    hip.neuron_fwd(xa, s1, D, B, hw * Cc, D * hw * Cc, hw * Cc, D * hw * Cc, hw * Cc, blk.sn1)
    s2 = torch.empty((ntok, Ch), dtype=torch.uint8, device=DEV)
    hip.spike_gemm_sn(s1, blk.fc1.Wp, s2, Ch, Cc, D, B * hw, hw, D * hw, hw, blk.sn2, alpha=blk.fc1.alpha, beta=blk.fc1.beta)
    hip.spike_gemm(s2, blk.fc2.Wp, xa, ntok, Cc, Ch, alpha=blk.fc2.alpha, beta=blk.fc2.beta, resid=xa)
    xb = hip.ms_mlp(x0.clone(), blk.fc1, blk.fc2, blk.sn1, blk.sn2, three_launches=True)   # (the one-launch form: tests/test_ms_mlp_fused_gpu.py)
    torch.cuda.synchronize()
    assert torch.equal(xa, xb) and not torch.equal(xa, x0)


@pytest.mark.parametrize("Tq,detach,v_reset", [(2, True, None), (2, False, None), (2, True, 0.0), (4, True, None), (1, True, None)])
def test_qk_gate_train_kernels_match_autograd_of_the_composed_expression(Tq, detach, v_reset):
    """sdf_qk_gate_f32_fwd / sdf_qk_gate_bwd against torch autograd through the reference's expression
    (`q.sum` per head -> neuron -> broadcast -> `k.mul`) with the CPU restatement of the neuron's backward: forward bit-equal
    (head sums of spikes are exact integers), gradients to fp32 summation order."""
    rows, Cc, nH = 4000 + 37, 96, 3
    q = (torch.rand((Tq, rows, Cc)) < 0.35).float()
    k = (torch.rand((Tq, rows, Cc)) < 0.3).float()
    ge = torch.randn((Tq, rows, Cc))
    p = hip.NeuronParams("lif", 2.0, 0.1 if v_reset is None else 3.0, v_reset)
    # reference on the CPU: explicit chain, neuron backward from the oracle's restatement
    s = q.reshape(Tq, rows, nH, 32).sum(-1)
    _, A = BW.lif_forward_h(s, p.tau, p.v_th, v_reset)
    e_ref = k * A.repeat_interleave(32, dim=-1)
    gk_ref = ge * A.repeat_interleave(32, dim=-1)
    gA = (ge * k).reshape(Tq, rows, nH, 32).sum(-1)
    gs = BW.lif_backward(s, gA, p.tau, p.v_th, v_reset, detach, 2.0)
    gq_ref = gs.repeat_interleave(32, dim=-1)
    e = hip.qk_gate_f32(q.to(DEV), k.to(DEV), p)
    gq, gk, _, _ = hip.qk_gate_bwd(q.to(DEV), k.to(DEV), ge.to(DEV), p, detach, 2.0)
    assert torch.equal(e.cpu(), e_ref) and torch.equal(gk.cpu(), gk_ref)
    assert (gq.cpu() - gq_ref).abs().max().item() <= 1e-5 * gq_ref.abs().max().item()
    # the autograd bridge
    from sdformerflow_amd.autograd import QKGateFunction
    qd, kd = q.to(DEV).requires_grad_(True), k.to(DEV).requires_grad_(True)
    QKGateFunction.apply(qd, kd, None, None, "lif", p.tau, p.v_th, v_reset, detach, 2.0).backward(ge.to(DEV))
    assert torch.equal(qd.grad, gq) and torch.equal(kd.grad, gk)


@pytest.mark.parametrize("Tq", [2, 4])
def test_qk_gate_train_kernels_psn_gate_with_parameter_gradients(Tq):
    """PSN gate (learnable T' x T' matrix + bias inside the gate): e, dL/dq, dL/dk, dL/dW, dL/db of the one-launch kernels against
    torch autograd through the composed expression with the reference's surrogate; parameter gradients deterministic."""
    from oracle import sdformer_oracle as O
    rows, Cc, nH = 3000 + 11, 96, 3
    g = torch.Generator().manual_seed(Tq)
    q = (torch.rand((Tq, rows, Cc), generator=g) < 0.35).float()
    k = (torch.rand((Tq, rows, Cc), generator=g) < 0.3).float()
    ge = torch.randn((Tq, rows, Cc), generator=g)
    W = (torch.randn((Tq, Tq), generator=g) * 0.3).requires_grad_(True)
    b = torch.full((Tq, 1), -2.0).requires_grad_(True)
    qr, kr = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
    s = qr.reshape(Tq, rows, nH, 32).sum(-1)
    h = torch.addmm(b, W, s.flatten(1))
    A = O._ATan.apply(h, 2.0).view(s.shape)
    e_ref = kr * A.repeat_interleave(32, dim=-1)
    e_ref.backward(ge)
    from sdformerflow_amd.autograd import QKGateFunction
    outs = []
    for _ in range(2):
        qd, kd = q.to(DEV).requires_grad_(True), k.to(DEV).requires_grad_(True)
        Wd, bd = W.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
        e = QKGateFunction.apply(qd, kd, Wd, bd, "psn", 2.0, 0.0, None, True, 2.0)
        e.backward(ge.to(DEV))
        outs.append((e.detach(), qd.grad, kd.grad, Wd.grad, bd.grad))
    assert all(torch.equal(a, c) for a, c in zip(*outs))
    e, gq, gk, gW, gb = (t.cpu() for t in outs[0])
    assert (e - e_ref.detach()).abs().mean().item() <= 1e-5                   # a head sum whose h is within rounding of 0 may flip
    for name, got, ref in (("gq", gq, qr.grad), ("gk", gk, kr.grad), ("gW", gW, W.grad), ("gb", gb, b.grad)):
        assert got.shape == ref.shape, name
        assert (got - ref).abs().max().item() <= 2e-4 * ref.abs().max().item() + 1e-7, (name, (got - ref).abs().max().item())


@pytest.mark.parametrize("R,Cc", [(5003, 96), (1237, 192), (40960, 384), (130, 3072), (7, 8)])
def test_channel_last_train_batchnorm_matches_fp64_autograd(R, Cc):
    """sdf_bn_train_fwd / _bwd (batch statistics down the rows of a channel-last matrix, fp64 sums) against the defining
    formulas evaluated in fp64 under autograd (what nn.BatchNorm2d in train mode computes on the reference's permuted view):
    output, running statistics, grad_x / grad_weight / grad_bias to 2e-6 of each tensor's scale; bit-equal run to run."""
    from sdformerflow_amd.autograd import BatchNormLastFunction
    g = torch.Generator().manual_seed(R + Cc)
    x0 = (torch.randn((R, Cc), generator=g) * 1.5 + 0.3).to(DEV)
    w0, b0 = (torch.rand(Cc, generator=g) + 0.5).to(DEV), torch.randn(Cc, generator=g).to(DEV)
    gy = torch.randn((R, Cc), generator=g).to(DEV)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x0, w0, b0))
    m, v = xd.mean(0), xd.var(0, unbiased=False)
    yd = (xd - m) / torch.sqrt(v + 1e-5) * wd + bd
    yd.backward(gy.double())
    ref = [yd.detach(), 0.1 * m.detach(), 0.9 + 0.1 * v.detach() * R / max(R - 1, 1), xd.grad, wd.grad, bd.grad]
    outs = []
    for _ in range(2):
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        rm, rv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
        y = BatchNormLastFunction.apply(x.view(1, R, Cc), w, b, rm, rv, 0.1, 1e-5).view(R, Cc)
        y.backward(gy)
        outs.append([t.detach().clone() for t in (y, rm, rv, x.grad, w.grad, b.grad)])
    for name, r, got, again in zip(("y", "running_mean", "running_var", "gx", "gw", "gb"), ref, *outs):
        assert torch.equal(got, again), name
        scale = r.abs().max().item() + 1e-12
        assert (got.double() - r).abs().max().item() <= 2e-6 * scale, (name, (got.double() - r).abs().max().item(), scale)


@pytest.mark.parametrize("N,Cc,H,W", [(20, 48, 36, 48), (3, 96, 9, 12), (5, 768, 6, 6), (2, 2, 4, 4)])
def test_nchw_train_batchnorm_matches_fp64_autograd(N, Cc, H, W):
    """sdf_bn_train_nchw_fwd / _bwd against the defining formulas in fp64 under autograd (nn.BatchNorm2d in train mode)."""
    from sdformerflow_amd.autograd import BatchNormNCHWFunction
    g = torch.Generator().manual_seed(N * 1000 + Cc)
    x0 = (torch.randn((N, Cc, H, W), generator=g) * 1.5 + 0.3).to(DEV)
    w0, b0 = (torch.rand(Cc, generator=g) + 0.5).to(DEV), torch.randn(Cc, generator=g).to(DEV)
    gy = torch.randn((N, Cc, H, W), generator=g).to(DEV)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x0, w0, b0))
    m, v = xd.mean((0, 2, 3)), xd.var((0, 2, 3), unbiased=False)
    yd = (xd - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + 1e-5) * wd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1)
    yd.backward(gy.double())
    R = N * H * W
    ref = [yd.detach(), 0.1 * m.detach(), 0.9 + 0.1 * v.detach() * R / (R - 1), xd.grad, wd.grad, bd.grad]
    outs = []
    for _ in range(2):
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        rm, rv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
        y = BatchNormNCHWFunction.apply(x, w, b, rm, rv, 0.1, 1e-5)
        y.backward(gy)
        outs.append([t.detach().clone() for t in (y, rm, rv, x.grad, w.grad, b.grad)])
    for name, r, got, again in zip(("y", "running_mean", "running_var", "gx", "gw", "gb"), ref, *outs):
        assert torch.equal(got, again), name
        scale = r.abs().max().item() + 1e-12
        assert (got.double() - r).abs().max().item() <= 2e-6 * scale, (name, (got.double() - r).abs().max().item(), scale)


@pytest.mark.parametrize("shape", [(2, 4, 18, 21, (2, 9, 9), (1, 4, 4)), (1, 10, 24, 24, (2, 8, 8), (0, 0, 0)), (1, 6, 15, 33, (2, 15, 15), (1, 7, 7))])
def test_window_gather_scatter_equal_the_reference_pad_roll_partition(shape):
    """Training-path window partition / reverse as row moves through the slice map (sdf_rows_gather_fwd / _scatter_fwd) against
    the reference's own sequence - F.pad, torch.roll(-shift), window_partition_v2's view / permute / raw view, and back
    (window_reverse, roll(+shift), crop) - bit-equal in both directions of autograd."""
    from sdformerflow_amd.autograd import WindowGatherFunction, WindowScatterFunction
    import torch.nn.functional as F
    B, D, H, W, ws, ss = shape
    Cc = 32
    Wd, Wh, Ww = ws
    x0 = torch.randn((B, D, H, W, Cc), device=DEV)
    gy = torch.randn((B, D, H, W, Cc), device=DEV)
    # reference composition
    xr = x0.clone().requires_grad_(True)
    pd, ph, pw = (Wd - D % Wd) % Wd, (Wh - H % Wh) % Wh, (Ww - W % Ww) % Ww
    xp = F.pad(xr, (0, 0, 0, pw, 0, ph, 0, pd))
    Dp, Hp, Wp = D + pd, H + ph, W + pw
    if any(ss):
        xp = torch.roll(xp, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
    xw = xp.view(B, Dp // Wd, Wd, Hp // Wh, Wh, Wp // Ww, Ww, Cc).permute(0, 1, 3, 5, 2, 4, 6, 7).contiguous()
    B_ = B * (Dp // Wd) * (Hp // Wh) * (Wp // Ww)
    win_ref = xw.view(Wd, B_, Wh * Ww, Cc)
    back = (win_ref * 2.0).reshape(B, Dp // Wd, Hp // Wh, Wp // Ww, Wd, Wh, Ww, Cc).permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, Dp, Hp, Wp, Cc)
    if any(ss):
        back = torch.roll(back, shifts=ss, dims=(1, 2, 3))
    y_ref = back[:, :D, :H, :W, :].contiguous()
    y_ref.backward(gy)
    # row moves
    xm = x0.clone().requires_grad_(True)
    row_map, B2 = hip.window_slice_map(B, D, H, W, ws, ss, DEV)
    win = WindowGatherFunction.apply(xm, row_map).view(Wd, B2, Wh * Ww, Cc)
    y = WindowScatterFunction.apply((win * 2.0).reshape(-1, Cc), row_map, (B, D, H, W, Cc))
    y.backward(gy)
    assert B2 == B_ and torch.equal(win, win_ref) and torch.equal(y, y_ref) and torch.equal(xm.grad, xr.grad)


@pytest.mark.parametrize("imgs,H,W,N,stride", [(3, 10, 14, 96, 2), (2, 9, 7, 192, 2), (1, 6, 5, 96, 1)])
def test_pointwise_conv_f32_exact_matrix_pipe(imgs, H, W, N, stride):
    """1x1 strided convolution of a real-valued membrane on the fp32 MFMA (SpikingPEDLayer.conv_res, Spiking_modules.py:819)
    against fp64: fp32 products and sums, so the error is that of an fp32 dot product of 96 terms."""
    x = rnd((imgs, H, W, 96), 140, -2.0, 2.0)
    w = rnd((N, 96), 141, -0.3, 0.3)
    b = rnd((N,), 142, -0.1, 0.1)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double().view(N, 96, 1, 1), b.double(), stride).permute(0, 2, 3, 1)
    got = hip.pointwise_conv_f32(x.to(DEV), w.to(DEV), stride, b.to(DEV)).cpu()
    assert got.shape == ref.shape
    assert (got.double() - ref).abs().max() <= 2e-6 * ref.abs().max()
    got0 = hip.pointwise_conv_f32(x.to(DEV), w.to(DEV), stride).cpu()
    assert (got0.double() - (ref - b.double())).abs().max() <= 2e-6 * ref.abs().max()
    assert not hip.pointwise_conv_supported(64, 96)


@pytest.mark.parametrize("T", [4, 10])
def test_plif_slttlif_glif_neurons_against_the_reference_fixture(T):
    """neuron types outside the shipped configurations (reference Spiking_modules.py:49-56, 75-92): plif runs on the LIF kernels through
    the multiplicative `tau` (0 < tau < 1, csrc/common.h sdf_inv_tau), SLTTlif IS the LIF forward - both bit-equal to the reference
    switch's output (tests/golden/neurons_extra.npz); glif is torch ops on the GPU tensor (its sigmoid gates may round differently on
    the device: decisions at the threshold may flip, nothing else)."""
    import numpy as np
    from sdformerflow_amd.STSwinNet_SNN.Spiking_modules import Spiking_neuron
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "neurons_extra.npz"))
    x = torch.from_numpy(g[f"x_T{T}"]).to(DEV)
    for tag, vr in (("soft", None), ("hard", 0.0), ("hard05", 0.05)):
        for kind in ("plif", "SLTTlif"):
            if f"{kind}_{tag}_T{T}_s" not in g:
                continue
            m = Spiking_neuron(num_steps=T, neuron_type=kind, v_th=0.1, v_reset=vr, tau=2.0).to(DEV).eval()
            if kind == "plif":
                m.load_state_dict({"spiking_neuron.w": torch.from_numpy(g[f"plif_T{T}_w"])})
                assert 0.0 < m.spiking_neuron.tau < 1.0
            want = torch.from_numpy(g[f"{kind}_{tag}_T{T}_s"]).to(DEV)
            assert torch.equal(m(x).to(torch.uint8), want), (kind, tag)
            # the strided descriptor form (what the engine's fused launches configure) takes the same parameters
            p = hip.NeuronParams("lif", m.spiking_neuron.tau, 0.1, vr)
            out = torch.empty(x.shape, dtype=torch.uint8, device=DEV)
            n = x[0].numel()
            hip.neuron_fwd(x, out, T, 1, n, 0, n, 0, n, p)
            assert torch.equal(out, want), (kind, tag)
    m = Spiking_neuron(num_steps=T, neuron_type="glif").to(DEV).eval()
    gsd = {kk[len(f"glif_T{T}/"):]: torch.from_numpy(v) for kk, v in g.items() if kk.startswith(f"glif_T{T}/")}
    m.load_state_dict(gsd)
    got, want = m(3.0 * x).to(torch.uint8), torch.from_numpy(g[f"glif_T{T}_s"]).to(DEV)
    assert (got != want).float().mean().item() <= 1e-3
    assert torch.equal(want.cpu(), O.glif_multistep(3.0 * x.cpu(), gsd, "spiking_neuron.").to(torch.uint8))      # the oracle is exact
    with pytest.raises(hip.SdfError):
        hip.lif_fwd(x, tau=1.0)                                                  # tau = 1 is neither form


@pytest.mark.parametrize("M,N,K", [(700, 96, 384), (4320, 1536, 384), (130, 192, 768)])
def test_spike_gemm_and_conv_take_sums_of_spikes_as_bytes(M, N, K):
    """The fp16-plane kernels expand ANY activation byte exactly (csrc/spike_mm.h expand_spikes: {n, 0x64} = fp16(1024 + n), minus
    1024): the SEW stream behind a patch merging - sums of spike tensors (reference Spiking_swin_transformer3D.py:840-845) - is their
    A operand as it stands.  Bytes 0..255 against fp64, GEMM and 3x3 convolution (the SEW res-block's first convolution)."""
    g = torch.Generator().manual_seed(M + K)
    a = torch.randint(0, 256, (M, K), generator=g, dtype=torch.uint8)
    a[::3] = torch.randint(0, 4, a[::3].shape, generator=g, dtype=torch.uint8)          # the realistic range as well
    w = rnd((N, K), 777, -0.1, 0.1)
    ref = a.double() @ w.double().t()
    out = torch.empty((M, N), device=DEV)
    hip.spike_gemm(a.to(DEV), hip.split_weight(w.to(DEV), 2), out, M, N, K)
    assert (out.cpu().double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    imgs, H, W, Cc = 3, 10, 12, 384
    x = torch.randint(0, 12, (imgs, H, W, Cc), generator=g, dtype=torch.uint8)
    wc = rnd((Cc, Cc, 3, 3), 778, -0.05, 0.05)
    cref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wc.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cc)
    cout = torch.empty((imgs * H * W, Cc), device=DEV)
    hip.spike_conv2d(x.to(DEV), hip.pack_conv_weight(wc.to(DEV), 2), imgs, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=cout)
    assert (cout.cpu().double() - cref).abs().max().item() <= 2e-6 * cref.abs().max().item()
