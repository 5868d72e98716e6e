"""The small-M 3x3 spike convolution of the U-Net bottleneck (csrc/ms_smallm.hip: one launch, K split over the waves of a workgroup,
exact integer reduction in LDS, BN + shortcut + neuron in the epilogue) through the C ABI `sdf_spike_conv2d_fwd` with int8 digit planes -
reference `MS_ResBlock.forward` (Spiking_modules.py:906-933: sn -> conv3x3 -> BN -> sn -> conv3x3 -> BN -> + identity):
  * fp32 epilogue (BN + shortcut): against an fp64 convolution with the weights the digit planes carry, to 1e-5 of the range;
  * fused neuron: the spikes are delta-consistent with the oracle neuron on that fp64 pre-activation (0 unexplained decisions), and
    bit-equal to the C oracle neuron applied to the kernel's OWN fp32 pre-activation when it also stores it (membrane form);
  * row-major digit planes and the same digits in fragment order (sdf_tile_weight_i8x3) give bit-equal results;
  * against the streaming kernel on the fp16 planes of the same weights (the A/B reference)."""
import pytest
import torch

from oracle import neuron_ref as R
from oracle import sdformer_oracle as O
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def spikes(shape, seed, rate=0.3):
    return (rnd(shape, seed) < rate).to(torch.uint8)


def _weff(dg):
    d = dg.cpu().double()
    return (d[2] * 65536 + d[1] * 256 + d[0]) * dg.sdf_col_scale.cpu().double().view(-1, 1)


def _ref(x, w_eff, Cin, Cout, alpha, beta, resid):
    w = w_eff.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)                  # planes are packed (ky, kx, cin)
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w, None, 1, 1).permute(0, 2, 3, 1).reshape(-1, Cout)
    return y * alpha.double() + beta.double() + (resid.double() if resid is not None else 0)


@pytest.mark.parametrize("B,T,H,W,Cin,Cout,with_res", [(1, 10, 9, 12, 768, 768, True), (1, 10, 9, 12, 768, 768, False), (2, 10, 5, 7, 384, 96, True),
                                                     (1, 20, 6, 5, 128, 64, True), (1, 10, 1, 1, 256, 32, False), (1, 10, 8, 10, 192, 96, True), (4, 10, 3, 3, 64, 32, True)])
def test_fp32_epilogue_against_fp64(B, T, H, W, Cin, Cout, with_res, monkeypatch):
    imgs = B * T
    assert hip.smallm_conv_applicable(imgs, H, W, Cin, Cout, 1, T)
    x = spikes((imgs, H, W, Cin), 300 + H)
    w = rnd((Cout, Cin, 3, 3), 301, -0.05, 0.05)
    alpha, beta = rnd((Cout,), 302, 0.5, 1.5), rnd((Cout,), 303, -0.2, 0.2)
    resid = rnd((imgs * H * W, Cout), 304) if with_res else None
    dg = hip.pack_conv_weight_i8x3(w.to(DEV))
    out = torch.full((imgs * H * W, Cout), float("nan"), device=DEV)
    hip.spike_conv2d(x.to(DEV), dg, imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=out, alpha=alpha.to(DEV), beta=beta.to(DEV),
                     resid=None if resid is None else resid.to(DEV))
    we = _weff(dg)                                                         # what the digits carry: within 2^-22 of the row's largest weight
    wrow = w.permute(0, 2, 3, 1).reshape(Cout, -1)
    assert ((we - wrow.double()).abs().max(dim=1).values <= 2.0 ** -22 * wrow.abs().max(dim=1).values.double()).all()
    ref = _ref(x, we, Cin, Cout, alpha, beta, resid)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    alt = torch.full_like(out, float("nan"))                               # the digits in fragment order (what the engine packs): bit-equal
    hip.spike_conv2d(x.to(DEV), hip.tile_weight_i8x3(dg), imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=alt, alpha=alpha.to(DEV),
                     beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV))
    assert torch.equal(alt, out)
    if Cout % 48 == 0:                                                     # the 48-column tile (two operand steps in flight): bit-equal
        monkeypatch.setenv("SDF_SMALLM_CB", "3")
        alt3 = torch.full_like(out, float("nan"))
        hip.spike_conv2d(x.to(DEV), hip.tile_weight_i8x3(dg), imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=alt3, alpha=alpha.to(DEV),
                         beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV))
        monkeypatch.delenv("SDF_SMALLM_CB")
        assert torch.equal(alt3, out)
    # the streaming kernel on the fp16 planes of the same weights
    if Cout % 96 == 0:
        monkeypatch.setenv("SDF_SMALLM", "0")
        old = torch.empty_like(out)
        hip.spike_conv2d(x.to(DEV), hip.pack_conv_weight(w.to(DEV), 2), imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=old,
                         alpha=alpha.to(DEV), beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV))
        assert (out - old).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("kind,v_reset", [("lif", None), ("lif", 0.0), ("if", None), ("psn", None)])
@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(1, 10, 9, 12, 768, 768), (2, 10, 4, 5, 384, 96), (1, 20, 6, 5, 128, 64)])
@pytest.mark.parametrize("membrane", [False, True])
@pytest.mark.parametrize("tile", ["2", "3"])
def test_fused_neuron_forms(B, T, H, W, Cin, Cout, membrane, kind, v_reset, tile, monkeypatch):
    if tile == "3" and (Cout % 48 or not membrane):
        pytest.skip("the 48-column tile reads digits in fragment order, 48 | Cout")
    monkeypatch.setenv("SDF_SMALLM_CB", tile)
    if kind not in ("lif", "psn") or v_reset is not None:
        if (H, W) != (9, 12):
            pytest.skip("the other neuron classes are covered on the shipped shape")
    imgs, hw = B * T, H * W
    x = spikes((imgs, H, W, Cin), 310 + H)
    w = rnd((Cout, Cin, 3, 3), 311, -0.05, 0.05)
    alpha, beta = rnd((Cout,), 312, 0.5, 1.5), rnd((Cout,), 313, -0.1, 0.3)
    resid = rnd((imgs * hw, Cout), 314, -0.3, 0.3) if membrane else None
    pw = pb = None
    sd = {}
    if kind == "psn":                    # the shipped neuron (reference Spiking_submodules.py:183-211): its T x T matrix and bias
        pw = (torch.eye(T) * 0.8 + rnd((T, T), 315, -0.15, 0.15)).contiguous()
        pb = (torch.full((T,), -0.1) + rnd((T,), 316, -0.03, 0.03)).contiguous()
        p = hip.NeuronParams("psn", psn_w=pw.to(DEV), psn_b=pb.to(DEV))
        sd = {"w.weight": pw, "w.bias": pb.view(-1, 1)}
    else:
        p = hip.NeuronParams(kind, 2.0, 0.1, v_reset)
    dg = hip.pack_conv_weight_i8x3(w.to(DEV), tiled=membrane)            # (both weight layouts take part)
    sp = torch.full((imgs * hw, Cout), 7, dtype=torch.uint8, device=DEV)
    m = torch.full((imgs * hw, Cout), float("nan"), device=DEV) if membrane else None
    hip.spike_conv2d(x.to(DEV), dg, imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=m, out_spike=sp, alpha=alpha.to(DEV),
                     beta=beta.to(DEV), resid=None if resid is None else resid.to(DEV), sn=p, sn_T=T, pos=(B * hw, hw, T * hw, hw))
    ref = _ref(x, _weff(hip.pack_conv_weight_i8x3(w.to(DEV))), Cin, Cout, alpha, beta, resid)
    got = sp.cpu().view(B, T, hw, Cout).permute(1, 0, 2, 3).float().contiguous()
    ht = ref.view(B, T, hw, Cout).permute(1, 0, 2, 3).float().contiguous()
    delta = 16 * 2.0 ** -23 * max(float(ht.pow(2).mean().sqrt()), 0.1)
    rep = O.delta_consistent(ht, got, O.NeuronCfg(kind, 0.1, v_reset, 2.0, T), sd, "w.", delta)
    assert rep["unexplained"] == 0, rep
    assert 0.02 < got.mean() < 0.98
    if membrane:
        assert (m.cpu().double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        own = R.neuron_ref(m.cpu().view(B, T, hw, Cout).permute(1, 0, 2, 3).contiguous(), kind, 2.0, 0.1, v_reset, psn_w=pw, psn_b=pb)
        assert torch.equal(own, got), "spikes are not the neuron of the stored membrane"


@pytest.mark.parametrize("M,N,K,extras", [(1080, 3456, 1536, False), (1080, 96, 1536, True), (200, 64, 128, True), (10, 32, 64, False), (4320, 192, 832, True)])
def test_plain_product_with_fp32_epilogue(M, N, K, extras):
    """`sdf_spike_gemm_fwd` with digits in fragment order (nsplit = SDF_PLANES_I8X3_TILED): the first decoder's stacked-tap GEMM
    (reference Spiking_modules.py:461-474 written as one product + col2im) - exact integer sums against fp64 to 1e-6 of the range, and
    against the streaming kernel on the fp16 planes of the same weights."""
    assert hip.smallm_gemm_applicable(M, N, K) or K < 1024
    A = spikes((M, K), 400 + N, 0.25)
    Wt = rnd((N, K), 401, -0.07, 0.07)
    dg = hip.tile_weight_i8x3(hip.split_weight_i8x3(Wt.to(DEV)))
    kw, ref = {}, None
    We = _weff(hip.split_weight_i8x3(Wt.to(DEV)))
    assert ((We - Wt.double()).abs().max(dim=1).values <= 2.0 ** -22 * Wt.abs().max(dim=1).values.double()).all()
    ref = A.double() @ We.t()
    if extras:
        alpha, beta, bias, resid = rnd((N,), 402, 0.5, 1.5), rnd((N,), 403, -0.2, 0.2), rnd((N,), 404, -0.1, 0.1), rnd((M, N), 405)
        kw = dict(alpha=alpha.to(DEV), beta=beta.to(DEV), bias=bias.to(DEV), resid=resid.to(DEV))
        ref = (ref + bias.double()) * alpha.double() + beta.double() + resid.double()
    out = torch.full((M, N), float("nan"), device=DEV)
    hip.spike_gemm(A.to(DEV), dg, out, M, N, K, **kw)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    old = torch.empty_like(out)
    hip.spike_gemm(A.to(DEV), hip.split_weight(Wt.to(DEV), 2), old, M, N, K, **kw)
    assert (out - old).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K,extras", [(17280, 864, 416, False), (4320, 1728, 800, True), (200, 64, 96, True), (10, 32, 32, False), (1230, 96, 1024, True)])
def test_plain_product_on_row_major_digits(M, N, K, extras):
    """`sdf_spike_gemm_fwd` with ROW-MAJOR digit planes (nsplit = SDF_PLANES_I8X3): the stacked-tap products of the middle decoder levels
    (reference Spiking_modules.py:461-474 as one GEMM + col2im) on the weight-resident row-loop kernel (csrc/ms_res.hip: whole-K digits in
    LDS, K % 64 != 0 included) - exact integer sums against fp64 to 1e-6 of the range, in place over a shortcut, and against the streaming
    kernel on the fp16 planes of the same weights."""
    assert hip.res_gemm_applicable(M, N, K)
    A = spikes((M, K), 420 + N, 0.25)
    Wt = rnd((N, K), 421, -0.07, 0.07)
    dg = hip.split_weight_i8x3(Wt.to(DEV))
    We = _weff(dg)
    assert ((We - Wt.double()).abs().max(dim=1).values <= 2.0 ** -22 * Wt.abs().max(dim=1).values.double()).all()
    ref = A.double() @ We.t()
    kw, resid = {}, None
    if extras:
        alpha, beta, bias, resid = rnd((N,), 422, 0.5, 1.5), rnd((N,), 423, -0.2, 0.2), rnd((N,), 424, -0.1, 0.1), rnd((M, N), 425)
        kw = dict(alpha=alpha.to(DEV), beta=beta.to(DEV), bias=bias.to(DEV))
        ref = (ref + bias.double()) * alpha.double() + beta.double() + resid.double()
    out = resid.to(DEV).clone() if extras else torch.full((M, N), float("nan"), device=DEV)
    hip.spike_gemm(A.to(DEV), dg, out, M, N, K, resid=out if extras else None, **kw)
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    if K % 32 == 0:
        old = torch.empty((M, N), device=DEV)
        hip.spike_gemm(A.to(DEV), hip.split_weight(Wt.to(DEV), 2), old, M, N, K, resid=resid.to(DEV) if extras else None, **kw)
        assert (out - old).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_plain_product_refuses_what_it_is_not_built_for():
    Wt = rnd((64, 128), 410, -0.1, 0.1)
    dg = hip.tile_weight_i8x3(hip.split_weight_i8x3(Wt.to(DEV)))
    out = torch.empty((12, 64), device=DEV)
    with pytest.raises(hip.SdfError):
        hip.spike_gemm(spikes((12, 128), 411).to(DEV), dg, out, 12, 64, 128)                 # M % 10 != 0
    with pytest.raises(hip.SdfError):
        hip.spike_gemm(spikes((12, 128), 412).to(DEV), hip.split_weight_i8x3(Wt.to(DEV)), torch.empty((12, 64), device=DEV), 12, 64, 128)   # row-major digits: rows in tens
