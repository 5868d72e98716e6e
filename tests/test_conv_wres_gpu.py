"""Weight-resident 3x3 spike convolution (csrc/spike_conv_wres.hip) against the streaming ping-pong kernel it replaces for
large launches (bit-equal: same planes, same k order inside one accumulator) and against fp64 / the C-oracle neuron."""
import os

import pytest
import torch

from oracle import neuron_ref as R
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def spikes(shape, seed, rate=0.3):
    return (rnd(shape, seed) < rate).to(torch.uint8)


def pack(w, ns):
    return hip.pack_conv_weight_i8x3(w.to(DEV)) if ns == "i8x3" else hip.pack_conv_weight(w.to(DEV), ns)


def both(fn):
    """fn() with the weight-resident kernel forced and with it disabled."""
    outs = []
    for mode in ("2", "0"):
        with hip.scoped_switches(SDF_CONV_WRES=mode):
            outs.append(fn())
    return outs


@pytest.mark.parametrize("ns", [1, 2, "i8x3"])
@pytest.mark.parametrize("imgs,H,W,Cout,with_res", [(10, 72, 96, 96, True), (3, 21, 37, 96, True), (2, 8, 16, 64, False), (5, 70, 90, 192, False)])
def test_fp32_epilogue_equals_streaming_kernel_and_fp64(imgs, H, W, Cout, with_res, ns):
    """Tile-aligned, ragged (21 x 37: partial tiles in both directions) and single-tile images; 64 / 96 / 192 output columns."""
    Cin = 96
    x = spikes((imgs, H, W, Cin), 200 + H)
    w = rnd((Cout, Cin, 3, 3), 201, -0.1, 0.1)
    alpha, beta = rnd((Cout,), 202, 0.5, 1.5), rnd((Cout,), 203, -0.2, 0.2)
    resid = rnd((imgs * H * W, Cout), 204) if with_res else None
    Wp = pack(w, ns)

    def run():
        out = torch.full((imgs * H * W, Cout), float("nan"), device=DEV)
        hip.spike_conv2d(x.to(DEV), Wp, imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=out, alpha=alpha.to(DEV),
                         beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV))
        return out.cpu()
    if ns == "i8x3" or Cout % 96:                                        # no streaming-kernel form (digit planes / N % 96 != 0)
        with hip.scoped_switches(SDF_CONV_WRES="2"):
            new = old = run()
    else:
        new, old = both(run)
    if Cout % 96 == 0 and ns != "i8x3":
        if imgs * H * W >= 60000:
            assert torch.equal(new, old)
        else:                                                            # small launches: the streaming path splits K (another summation order)
            assert (new - old).abs().max().item() <= 2e-6 * old.abs().max().item()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    ref = ref * alpha.double() + beta.double() + (resid.double() if with_res else 0)
    tol = 6e-3 if ns == 1 else 1e-5                                      # one bf16 plane carries 8 significand bits
    assert (new.double() - ref).abs().max().item() <= tol * ref.abs().max().item()


@pytest.mark.parametrize("ns", [1, 2, "i8x3"])
@pytest.mark.parametrize("kind,v_reset", [("lif", None), ("lif", 0.0), ("if", None)])
@pytest.mark.parametrize("order", ["tb", "bt"])
def test_fused_neuron_T10_equals_streaming_kernel_and_oracle(kind, v_reset, order, ns):
    """conv -> BN -> neuron over T = 10, time as the kernel's loop with the membrane in registers; both row orders the
    callers use ((t, b) images: the kernel tests; (b, t) images: the engine); spikes = C-oracle neuron of the fp64 conv except
    at threshold-rounding distance; bit-equal to the streaming kernel's fused epilogue."""
    T, B, H, W, Cc = 10, 2, 24, 32, 96
    x = spikes((T * B, H, W, Cc), 210)                                   # image index: t * B + b ("tb") or b * T + t ("bt")
    w = rnd((Cc, Cc, 3, 3), 211, -0.1, 0.1)
    alpha, beta = rnd((Cc,), 212, 0.5, 1.5), rnd((Cc,), 213, -0.2, 0.2)
    Wp = pack(w, ns)
    n = H * W
    pos = (B * n, B * n, 0, B * n) if order == "tb" else (B * n, n, T * n, n)
    sn = hip.NeuronParams(kind, 2.0, 0.1, v_reset)

    def run():
        out = torch.zeros((T * B * n, Cc), dtype=torch.uint8, device=DEV)
        hip.spike_conv2d(x.to(DEV), Wp, T * B, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out_spike=out, alpha=alpha.to(DEV),
                         beta=beta.to(DEV), sn=sn, sn_T=T, pos=pos)
        return out.cpu()
    if ns == "i8x3":
        new = run()
        assert torch.equal(new, run())                                   # exact integer sums: run-to-run bit-equal
    else:
        new, old = both(run)
        assert torch.equal(new, old)
    h = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1)
    h = (h * alpha.double() + beta.double()).float()                      # (T*B, H, W, C) in image order
    h = h.view(T, B, -1) if order == "tb" else h.view(B, T, -1).permute(1, 0, 2).contiguous()
    ref = R.neuron_ref(h.reshape(T, -1), kind, 2.0, 0.1, v_reset).view(T, B, -1)
    got = new.view(T, B, -1) if order == "tb" else new.view(B, T, -1).permute(1, 0, 2)
    if ns != 1:
        assert (got.float() != ref).float().mean().item() <= 2e-4
    assert 0.03 < ref.mean() < 0.97


@pytest.mark.parametrize("ns", [2, "i8x3"])
@pytest.mark.parametrize("with_resid", [True, False])
def test_fused_neuron_with_membrane_output(with_resid, ns):
    """MS_ResBlock conv2 + the next sn1 (Spiking_modules.py:922-933): fp32 membrane = BN(conv) + identity within 1e-5 of fp64,
    spikes = C-oracle neuron of the kernel's OWN membrane bit for bit, both equal to the streaming kernel."""
    T, B, H, W, Cc = 10, 1, 36, 48, 96
    x = spikes((T * B, H, W, Cc), 220)
    w = rnd((Cc, Cc, 3, 3), 221, -0.1, 0.1)
    alpha, beta = rnd((Cc,), 222, 0.5, 1.5), rnd((Cc,), 223, -0.2, 0.2)
    resid = rnd((T * B * H * W, Cc), 224) if with_resid else None
    n = B * H * W
    Wp = pack(w, ns)

    def run():
        m = torch.full((T * n, Cc), float("nan"), device=DEV)
        sp = torch.zeros((T * n, Cc), dtype=torch.uint8, device=DEV)
        hip.spike_conv2d(x.to(DEV), Wp, T * B, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=m, out_spike=sp,
                         alpha=alpha.to(DEV), beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV),
                         sn=hip.NeuronParams("lif", 2.0, 0.1, None), sn_T=T, pos=(n, n, 0, n))
        return m.cpu(), sp.cpu()
    if ns == "i8x3":
        m, sp = run()
    else:
        (m, sp), (m0, sp0) = both(run)
        assert torch.equal(m, m0) and torch.equal(sp, sp0)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cc)
    ref = ref * alpha.double() + beta.double() + (resid.double() if with_resid else 0)
    assert (m.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert torch.equal(sp.float(), R.neuron_ref(m.view(T, -1), "lif", 2.0, 0.1, None).view(T * n, Cc))


@pytest.mark.parametrize("T,B,H,W", [(20, 2, 24, 32), (5, 2, 24, 48), (20, 4, 32, 32)])
def test_digit_kernel_fused_neuron_at_T5_and_T20(T, B, H, W):
    """The digit kernel's fused form rolls its time loop: T = 5 / 20 (BASELINE configs[4]: 20 bins) run the same kernel as T = 10.
    (b, t) image order as the engine uses it; fp32 membrane within 1e-5 of fp64, spikes = C-oracle neuron of the kernel's OWN
    membrane bit for bit; the spikes-only form gives the same spikes."""
    Cc = 96
    x = spikes((B * T, H, W, Cc), 230 + T)
    w = rnd((Cc, Cc, 3, 3), 231, -0.1, 0.1)
    alpha, beta = rnd((Cc,), 232, 0.5, 1.5), rnd((Cc,), 233, -0.2, 0.2)
    n = H * W
    resid = rnd((B * T * n, Cc), 234)
    Wp = pack(w, "i8x3")
    pos = (B * n, n, T * n, n)
    sn = hip.NeuronParams("lif", 2.0, 0.1, None)
    m = torch.full((B * T * n, Cc), float("nan"), device=DEV)
    sp = torch.zeros((B * T * n, Cc), dtype=torch.uint8, device=DEV)
    hip.spike_conv2d(x.to(DEV), Wp, B * T, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=m, out_spike=sp, alpha=alpha.to(DEV),
                     beta=beta.to(DEV), resid=resid.to(DEV), sn=sn, sn_T=T, pos=pos)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cc)
    ref = ref * alpha.double() + beta.double() + resid.double()
    m = m.cpu()
    assert (m.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    mt = m.view(B, T, n * Cc).permute(1, 0, 2).contiguous()                                     # (T, B, ...)
    want = R.neuron_ref(mt.view(T, -1), "lif", 2.0, 0.1, None).view(T, B, n * Cc).permute(1, 0, 2).reshape(B * T * n, Cc)
    assert torch.equal(sp.cpu().float(), want) and 0.03 < want.mean() < 0.97
    sp2 = torch.zeros_like(sp)
    hip.spike_conv2d(x.to(DEV), Wp, B * T, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out_spike=sp2, alpha=alpha.to(DEV),
                     beta=beta.to(DEV), sn=sn, sn_T=T, pos=pos)
    h0 = (torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cc)
          * alpha.double() + beta.double()).float()
    w0 = R.neuron_ref(h0.view(B, T, n * Cc).permute(1, 0, 2).contiguous().view(T, -1), "lif", 2.0, 0.1, None)
    w0 = w0.view(T, B, n * Cc).permute(1, 0, 2).reshape(B * T * n, Cc)
    assert (sp2.cpu().float() != w0).float().mean().item() <= 2e-4


@pytest.mark.parametrize("Cin", [48, 96])
@pytest.mark.parametrize("imgs,H,W,Cout,with_res", [(10, 48, 64, 96, True), (3, 21, 37, 96, True), (2, 16, 32, 64, False), (4, 35, 66, 32, False),
                                                    (1, 1, 1, 96, False)])
def test_stride2_on_48_channels_fp32_epilogue(imgs, H, W, Cout, with_res, Cin):
    """The patch embedding's first 3x3 (48 -> 96, stride 2, pad 1: Spiking_modules.py:1776-1779) on the digit kernel's stride-2 form
    (17 x 33 halo with the even / odd columns as two planes, K order permuted so that a K step's two pieces sit a constant distance
    apart): even, odd and ragged images - the last input row / column is a tap of the last output only when H / W is even.
    Cin = 96 (the projection 96 -> C at stride 2, :1786-1789): the same halo form in two channel passes of 48."""
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = spikes((imgs, H, W, Cin), 300 + H)
    w = rnd((Cout, Cin, 3, 3), 301, -0.1, 0.1)
    alpha, beta = rnd((Cout,), 302, 0.5, 1.5), rnd((Cout,), 303, -0.2, 0.2)
    resid = rnd((imgs * OH * OW, Cout), 304) if with_res else None
    Wp = pack(w, "i8x3")
    out = torch.full((imgs * OH * OW, Cout), float("nan"), device=DEV)
    hip.spike_conv2d(x.to(DEV), Wp, imgs, H, W, Cin, OH, OW, 3, 3, 2, (-1, 0, 1), (-1, 0, 1), out=out, alpha=alpha.to(DEV),
                     beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 2, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    ref = ref * alpha.double() + beta.double() + (resid.double() if with_res else 0)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # and bit-equal to the 16-bit-plane streaming kernel's result where the digits and the planes hold the same weight: a weight
    # that is a multiple of 2^-10 below 1/8 is exact in both
    wq = (w * 1024).round() / 1024
    if Cout % 96 == 0:                                                  # (the streaming kernels work on 96-column blocks)
        o2 = []
        for P in (pack(wq, "i8x3"), pack(wq, 2)):
            o2.append(torch.full((imgs * OH * OW, Cout), float("nan"), device=DEV))
            hip.spike_conv2d(x.to(DEV), P, imgs, H, W, Cin, OH, OW, 3, 3, 2, (-1, 0, 1), (-1, 0, 1), out=o2[-1])
        assert torch.equal(o2[0], o2[1])                                # (sums of multiples of 2^-10 below 2^6: exact in fp32 whatever the order)


@pytest.mark.parametrize("kind,v_reset", [("lif", None), ("lif", 0.0), ("if", None)])
@pytest.mark.parametrize("T,B,H,W", [(10, 1, 48, 64), (10, 2, 21, 37), (20, 2, 24, 32), (5, 1, 40, 48)])
def test_stride2_on_48_channels_fused_neuron(T, B, H, W, kind, v_reset):
    """conv (stride 2) -> BN -> neuron over T with the membrane handed on as well ((b, t) images, as the engine calls it): fp32
    membrane within 1e-5 of fp64, spikes = C-oracle neuron of the kernel's OWN membrane bit for bit; the spikes-only form gives the
    same spikes bit for bit (same arithmetic, nothing stored)."""
    Cin, Cc = 48, 96
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = spikes((B * T, H, W, Cin), 310 + T)
    w = rnd((Cc, Cin, 3, 3), 311, -0.15, 0.15)
    alpha, beta = rnd((Cc,), 312, 0.5, 1.5), rnd((Cc,), 313, -0.2, 0.2)
    n = OH * OW
    Wp = pack(w, "i8x3")
    pos = (B * n, n, T * n, n)
    sn = hip.NeuronParams(kind, 2.0, 0.1, v_reset)
    m = torch.full((B * T * n, Cc), float("nan"), device=DEV)
    sp = torch.zeros((B * T * n, Cc), dtype=torch.uint8, device=DEV)
    hip.spike_conv2d(x.to(DEV), Wp, B * T, H, W, Cin, OH, OW, 3, 3, 2, (-1, 0, 1), (-1, 0, 1), out=m, out_spike=sp, alpha=alpha.to(DEV),
                     beta=beta.to(DEV), sn=sn, sn_T=T, pos=pos)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 2, 1).permute(0, 2, 3, 1).reshape(-1, Cc)
    ref = ref * alpha.double() + beta.double()
    m = m.cpu()
    assert (m.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    mt = m.view(B, T, n * Cc).permute(1, 0, 2).contiguous()
    want = R.neuron_ref(mt.view(T, -1), kind, 2.0, 0.1, v_reset).view(T, B, n * Cc).permute(1, 0, 2).reshape(B * T * n, Cc)
    assert torch.equal(sp.cpu().float(), want) and 0.03 < want.mean() < 0.97
    sp2 = torch.zeros_like(sp)
    hip.spike_conv2d(x.to(DEV), Wp, B * T, H, W, Cin, OH, OW, 3, 3, 2, (-1, 0, 1), (-1, 0, 1), out_spike=sp2, alpha=alpha.to(DEV),
                     beta=beta.to(DEV), sn=sn, sn_T=T, pos=pos)
    assert torch.equal(sp2, sp)


def test_i8x3_digit_planes_reconstruct_the_weight():
    """sdf_split_weight_i8x3: (d2*65536 + d1*256 + d0) * scale is the weight to 2^-23 of the row maximum, digits in range."""
    w = rnd((96, 864), 230, -0.2, 0.2)
    w[5] = 0.0                                                           # an all-zero row
    w[7, 3] = 0.19999999                                                 # the row maximum
    planes = hip.split_weight_i8x3(w.to(DEV))
    d = planes.cpu().to(torch.float64)
    sc = planes.sdf_col_scale.cpu().to(torch.float64)
    assert planes.dtype == torch.int8 and d[2].abs().max() <= 127       # (23 bits against 2^ceil(log2 max) where the digits hold them)
    rec = (d[2] * 65536 + d[1] * 256 + d[0]) * sc[:, None]
    err = (rec - w.double()).abs().max(1).values
    assert (err <= w.double().abs().max(1).values * 2.0 ** -22 + 1e-300).all()
    assert (torch.log2(sc) == torch.log2(sc).round()).all()             # powers of two
