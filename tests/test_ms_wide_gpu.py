"""GPU parity of the digit-plane kernels of the MS swin block - csrc/ms_wide.hip (swin stages 2 / 3, C = 384 / 768) and, since round 5,
csrc/ms_res.hip (stages 0 / 1 and the first merge, C = 96 / 192: whole-K weights LDS-resident, row loop, two waves per SIMD) - through the C ABI
`sdf_ms_mlp_fwd` / `sdf_qk_attn_fwd`, STEP BY STEP against the oracle - every step on the kernels' OWN upstream spikes
(teacher forcing, as tests/replay.py does for whole models), so that each statement is exact:

MLP (reference Spiking_swin_transformer3D.py:164-181, :845)
  * SN1 spikes: bit-equal to the C oracle neuron on x;
  * SN2 spikes: delta-consistent with the oracle neuron on the fp64 pre-activation BN1(s1 W1^T) - 0 decisions that the
    reference's own threshold margin (16 ulp of max(rms, v_th)) does not explain;
  * output: x + BN2(s2 W2^T) in fp64, to 1e-5 of the output range.
Attention (reference :661-717, :781-821, :840)
  * slice spikes SN_proj(x gathered through the slice map): bit-equal to the oracle neuron;
  * q | k spikes: delta-consistent with the oracle neuron on the fp64 BN(xs W^T) (+ positional term on k);
  * token gate: E == k AND SN2_q(head sums of q), an integer statement - exact;
  * output: x + scatter(BN(Z Wp^T + b)), Z = E through the reference's raw head reshape, in fp64 to 1e-5;
  * the emitted first neuron of the MLP: bit-equal to the oracle neuron on the kernel's own updated x.
And both against the general kernels of the same entry points (flag NARROW: the A/B reference)."""
import pytest
import torch

from oracle import neuron_ref as R
from oracle import sdformer_oracle as O
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True, params=["row-loop", "k-ring"])
def _kernel_family(request, monkeypatch):
    """Two kernel families serve these entry points: the weight-resident row-loop kernels (csrc/ms_res.hip: any K <= 1024 in steps of 16;
    by default every stage from 128 channels on - a chip-time choice measured on the headline, profiles/r5o_routing_ab.txt) and the
    K-ring kernels (csrc/ms_wide.hip: C >= 192 in steps of 64, and whatever the row loop refuses - fc2 of the stages with 4 C > 1024).
    Every test runs on both: "row-loop" = from 64 channels on, "k-ring" = the row loop below 192 channels only."""
    monkeypatch.setenv("SDF_RES_MINC", "64")
    if request.param == "k-ring":
        monkeypatch.setenv("SDF_RES_MAXC", "191")
    return request.param


class _L:
    def __init__(self, W, alpha, beta, bias=None, ns=2):
        self.N, self.K = W.shape
        self.Wp = hip.split_weight(W.to(DEV).contiguous(), ns)
        self.digits = hip.split_weight_i8x3(W.to(DEV).contiguous())
        if self.K % 64 == 0 and self.N % 16 == 0:
            self.digits_tiled = hip.tile_weight_i8x3(self.digits)        # (fc2 of <= 5 120 tokens takes the small-M kernel through these)
        self.alpha, self.beta = alpha.to(DEV).contiguous(), beta.to(DEV).contiguous()
        self.bias = None if bias is None else bias.to(DEV).contiguous()


def _weff(dg):
    """The fp64 value of the weight the int8 digit planes carry (what the wide-stage kernels multiply by, exactly)."""
    d = dg.cpu().double()
    return (d[2] * 65536 + d[1] * 256 + d[0]) * dg.sdf_col_scale.cpu().double().view(-1, 1)


def _weff16(Wp):
    """The fp64 value of the weight the two fp16 planes carry, (hi + lo) / scale (sdf_split_weight_f16x2): what the general kernels of
    the narrowest stage (csrc/qk_front.hip, spike_gemm.hip, ms_mlp_fused.hip) multiply by."""
    h = Wp.cpu().view(torch.float16).double()
    return (h[0] + h[1]) * float(Wp.sdf_acc_scale)


NEURONS = {                        # name -> (kind, tau, v_reset): class 0 (soft reset, power-of-two tau / plif), the general class 2,
    "lif": ("lif", 2.0, None),     # and class 1 = the PSN the reference ships (configs/train_DSEC_supervised_SDformerFlow_en4.yml:49)
    "plif": ("lif", 0.3775406777858734, None),
    "lif_hard": ("lif", 2.0, 0.0),
    "if": ("if", 2.0, None),
    "psn": ("psn", 2.0, None),
}
ALL = ["lif", "plif", "lif_hard", "if", "psn"]


class _N:
    """One neuron instance for both sides: NeuronParams for the kernels, the C oracle neuron, the NeuronCfg / state dict of
    O.delta_consistent.  PSN (reference Spiking_submodules.py:183-211): its own T x T matrix and bias per instance - `gain` scales
    the matrix to the size of the pre-activation (the token gate sees head sums 0..32), `bias` sets the firing rate."""

    def __init__(self, name, T, v_th=0.1, seed=0, gain=1.0, bias=-0.1):
        self.name, self.T, self.v_th = name, T, v_th
        self.kind, self.tau, self.vr = NEURONS[name]
        self.w = self.b = None
        if self.kind == "psn":
            self.w = ((torch.eye(T) * 0.8 + rnd((T, T), 7700 + seed, -0.15, 0.15)) * gain).contiguous()
            self.b = (torch.full((T,), float(bias)) + rnd((T,), 7800 + seed, -0.03, 0.03)).contiguous()
            self.p = hip.NeuronParams("psn", psn_w=self.w.to(DEV), psn_b=self.b.to(DEV))
        else:
            self.p = hip.NeuronParams(self.kind, self.tau, v_th, self.vr)

    def ref(self, xt):
        """spikes of the oracle neuron on xt (T, ...) fp32 (C restatement: true fmaf chain / separately rounded LIF ops)"""
        if self.kind == "psn":
            return R.neuron_ref(xt, "psn", psn_w=self.w, psn_b=self.b)
        return R.neuron_ref(xt, self.kind, self.tau, self.v_th, self.vr)

    def check(self, h, got):
        """O.delta_consistent of `got` against the pre-activation h (T, ...)"""
        sd = {"w.weight": self.w, "w.bias": self.b.view(-1, 1)} if self.kind == "psn" else {}
        cfg = O.NeuronCfg(self.kind, self.v_th, self.vr, self.tau, self.T)
        return O.delta_consistent(h, got, cfg, sd, "w.", _delta(h, self.v_th))


def _delta(h, v_th=0.1):
    return 16 * 2.0 ** -23 * max(float(h.pow(2).mean().sqrt()), v_th)


# ------------------------------------------------------------------------------------------------------------------ MLP
@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("B,D,H,W,Cc", [(1, 10, 18, 24, 384), (1, 10, 9, 12, 768), (2, 10, 5, 7, 384), (1, 20, 6, 5, 384),
                                        (1, 20, 3, 3, 768), (3, 10, 1, 3, 256), (1, 10, 36, 48, 192), (2, 20, 5, 3, 192), (1, 10, 4, 6, 320),
                                        # narrow stages (ms_res.hip): stage 0 at full size (K = 96: one and a half 64-deep steps), ragged units,
                                        # T = 20, K = 64 / 128 / 160
                                        (1, 10, 72, 96, 96), (2, 10, 5, 7, 96), (1, 20, 6, 5, 96), (3, 10, 1, 3, 64), (1, 10, 9, 7, 128), (1, 10, 4, 6, 160)])
def test_wide_mlp_steps_against_the_oracle(B, D, H, W, Cc, name):
    if name not in ("lif", "psn") and (D != 10 or B != 1 or H * W > 1000):
        pytest.skip("the other neuron classes are covered on the shipped T = 10 shapes")
    Ch, ntok = 4 * Cc, B * D * H * W
    x0 = rnd((B, D, H, W, Cc), 700, -0.5, 1.0)
    W1, W2 = rnd((Ch, Cc), 701, -0.15, 0.15), rnd((Cc, Ch), 702, -0.05, 0.05)
    a1, b1 = rnd((Ch,), 703, 0.5, 1.5), rnd((Ch,), 704, -0.2, 0.2)
    a2, b2 = rnd((Cc,), 705, 0.5, 1.5), rnd((Cc,), 706, -0.2, 0.2)
    fc1, fc2, n1, n2 = _L(W1, a1, b1), _L(W2, a2, b2), _N(name, D, seed=1), _N(name, D, seed=2)
    keep = []
    xg = hip.ms_mlp(x0.to(DEV).clone(), fc1, fc2, n1.p, n2.p, keep_ws=keep)
    torch.cuda.synchronize()
    ws = keep[0].cpu()
    s1g = ws[:ntok * Cc].view(ntok, Cc)
    s2g = ws[(ntok * Cc + 255) // 256 * 256:][:ntok * Ch].view(ntok, Ch)
    # (a) SN1 over D: bit-exact
    xt = x0.permute(1, 0, 2, 3, 4).contiguous()
    s1r = n1.ref(xt).permute(1, 0, 2, 3, 4).reshape(ntok, Cc)
    assert torch.equal(s1g.float(), s1r), "SN1 spikes differ from the oracle"
    assert 0.03 < s1r.mean() < 0.97
    # (b) SN2 on the kernel's own s1
    W1e, W2e = _weff(fc1.digits), _weff(fc2.digits)
    assert (W1e - W1.double()).abs().max() <= 2.0 ** -22 * W1.abs().max()
    h = (s1g.double() @ W1e.t()) * a1.double() + b1.double()
    ht = h.view(B, D, H * W, Ch).permute(1, 0, 2, 3).float().contiguous()
    got = s2g.view(B, D, H * W, Ch).permute(1, 0, 2, 3).float().contiguous()
    rep = n2.check(ht, got)
    assert rep["unexplained"] == 0, rep
    assert rep["flips"] <= 2e-4 * got.numel(), rep
    assert 0.03 < got.mean() < 0.97
    # (c) fc2 + BN2 + shortcut on the kernel's own s2
    ref = x0.reshape(ntok, Cc).double() + (s2g.double() @ W2e.t()) * a2.double() + b2.double()
    err = (xg.cpu().reshape(ntok, Cc).double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item(), err
    # (d) the general kernels of the same entry point: SN1 bit-equal, SN2 equal up to near-threshold decisions
    kb = []
    xb = hip.ms_mlp(x0.to(DEV).clone(), fc1, fc2, n1.p, n2.p, keep_ws=kb, narrow=True)
    torch.cuda.synchronize()
    wb = kb[0].cpu()
    assert torch.equal(wb[:ntok * Cc], ws[:ntok * Cc])
    s2b = wb[(ntok * Cc + 255) // 256 * 256:][:ntok * Ch].view(ntok, Ch)
    diff = s2b != s2g
    assert diff.float().mean().item() <= 1e-4
    same = ~diff.any(dim=1)
    assert (xg.cpu().view(ntok, Cc) - xb.cpu().view(ntok, Cc)).abs()[same].max().item() <= 2e-5 * ref.abs().max().item()


# ------------------------------------------------------------------------------------------------------------ attention
def _attn_case(B, D, H, W, Cc, window, shift, seed=0):
    nH, Tq, N1 = Cc // 32, window[0], window[1] * window[2]
    x0 = rnd((B, D, H, W, Cc), 800 + seed, -0.5, 1.0)
    Wq, Wk, Wp = (rnd((Cc, Cc), 801 + i + seed, -0.12, 0.12) for i in range(3))
    aq, bq = rnd((Cc,), 811 + seed, 0.5, 1.5), rnd((Cc,), 812 + seed, -0.1, 0.3)
    ak, bk = rnd((Cc,), 813 + seed, 0.5, 1.5), rnd((Cc,), 814 + seed, -0.1, 0.3)
    ap, bp, biasp = rnd((Cc,), 815 + seed, 0.5, 1.5), rnd((Cc,), 816 + seed, -0.2, 0.2), rnd((Cc,), 817 + seed, -0.1, 0.1)
    pe = rnd((Tq * N1, Cc), 818 + seed, -0.2, 0.2)
    return nH, Tq, N1, x0, Wq, Wk, Wp, aq, bq, ak, bk, ap, bp, biasp, pe


@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("stacked", [True, False])
@pytest.mark.parametrize("B,D,H,W,Cc,window,shift", [
    (1, 10, 36, 48, 192, (2, 9, 9), (1, 4, 4)),          # stage 1: K = 192 = one and a half 128-deep chunks
    (1, 10, 18, 24, 384, (2, 9, 9), (1, 4, 4)),          # stage 2 of the shipped model: padded width, shifted
    (1, 10, 18, 24, 384, (2, 9, 9), (0, 0, 0)),
    (1, 10, 9, 12, 768, (2, 9, 9), (1, 4, 4)),           # stage 3
    (1, 10, 9, 12, 640, (2, 9, 9), (1, 4, 4)),           # K = 5 chunks (odd: the main loop's single-chunk tail)
    (1, 10, 9, 12, 512, (2, 9, 9), (0, 0, 0)),           # K = 4 chunks
    (2, 10, 8, 11, 384, (2, 8, 8), (0, 0, 0)),           # batch 2, 64-token windows, ragged map
    (1, 20, 7, 9, 256, (2, 5, 5), (1, 2, 2)),            # T = 20, 25-token windows
    # narrow stages (ms_res.hip)
    (1, 10, 72, 96, 96, (2, 9, 9), (1, 4, 4)),           # stage 0 of the shipped model at full size: K = 96, padded width, shifted
    (1, 10, 36, 48, 96, (2, 9, 9), (1, 4, 4)),
    (2, 10, 8, 11, 96, (2, 8, 8), (0, 0, 0)),            # batch 2, ragged map
    (1, 20, 7, 9, 64, (2, 5, 5), (1, 2, 2)),             # T = 20, K = 64
    (1, 10, 18, 24, 160, (2, 9, 9), (1, 4, 4)),          # K = 160: two and a half steps
])
def test_wide_attention_steps_against_the_oracle(B, D, H, W, Cc, window, shift, stacked, name):
    _attention_steps(B, D, H, W, Cc, window, shift, stacked, name)


@pytest.mark.parametrize("name,stacked", [("lif", True), ("psn", False), ("plif", True)])
@pytest.mark.parametrize("B,D,H,W,Cc,window,shift", [(1, 10, 72, 96, 96, (2, 9, 9), (1, 4, 4)), (2, 10, 8, 11, 96, (2, 8, 8), (0, 0, 0)),
                                                     (1, 10, 36, 48, 192, (2, 9, 9), (1, 4, 4))])
def test_general_attention_steps_against_the_oracle(B, D, H, W, Cc, window, shift, stacked, name, _kernel_family):
    """The kernels stage 0 of the shipped model still runs on (one-launch front csrc/qk_front.hip + projection csrc/spike_gemm.hip on the
    fp16 planes: flag NARROW of the same entry point) held to the ORACLE step by step - slice spikes bit-equal, q | k delta-consistent,
    token gate exact, output to 1e-5 - not only to their own multi-launch form (tests/test_qk_front_gpu.py)."""
    if _kernel_family != "row-loop":
        pytest.skip("one kernel family here: the general kernels")
    _attention_steps(B, D, H, W, Cc, window, shift, stacked, name, general=True)


def _attention_steps(B, D, H, W, Cc, window, shift, stacked, name, general=False):
    if general:
        pass                                                      # (the caller's parametrisation is the selection)
    elif name == "psn" and stacked:
        pytest.skip("every PSN has its own matrix: q and k are separate projections (engine.py:_Block)")
    elif name == "psn":
        if B != 1 and D != 20:
            pytest.skip("PSN: the shipped shapes, every K chunk count and T = 20")
    elif (name != "lif" or not stacked) and ((Cc, H) not in ((384, 18), (96, 36)) or D != 10 or B != 1 or shift[0] != 1):
        pytest.skip("neuron classes / separate projections are covered on the shipped stage-2 shape and on a stage-0 shape")
    nH, Tq, N1, x0, Wq, Wk, Wp, aq, bq, ak, bk, ap, bp, biasp, pe = _attn_case(B, D, H, W, Cc, window, shift)
    npj, nq, nk_, ng, ne = _N(name, Tq, seed=11), _N(name, Tq, seed=12), _N(name, Tq, seed=13), \
        _N(name, Tq, seed=14, gain=0.12, bias=-1.2), _N(name, D, seed=15)
    plin = _L(Wp, ap, bp, biasp)
    qlin, klin = _L(Wq, aq, bq), _L(Wk, ak, bk)
    if stacked:
        wcat = torch.cat([Wq, Wk], 0).to(DEV).contiguous()
        qk = {"Wp": hip.split_weight(wcat, 2), "digits": hip.split_weight_i8x3(wcat), "alpha": torch.cat([aq, ak]).to(DEV),
              "beta": torch.cat([bq, bk]).to(DEV), "add": torch.cat([torch.zeros_like(pe), pe], -1).to(DEV).contiguous()}
        kw = dict(qk=qk)
        Wqe, Wke = (_weff16(qk["Wp"])[:Cc], _weff16(qk["Wp"])[Cc:]) if general else (_weff(qk["digits"])[:Cc], _weff(qk["digits"])[Cc:])
    else:
        kw = dict(q_lin=qlin, k_lin=klin, pe=pe.to(DEV).contiguous())
        Wqe, Wke = (_weff16(qlin.Wp), _weff16(klin.Wp)) if general else (_weff(qlin.digits), _weff(klin.digits))
    rowmap, B_ = hip.window_slice_map(B, D, H, W, window, shift, DEV)
    x_rows, rows = B * D * H * W, B_ * N1
    M = Tq * rows
    zsrc = hip.window_zsrc_map(rowmap, B_, Tq, N1, nH, x_rows)
    ws_mlp = torch.zeros((x_rows * Cc,), dtype=torch.uint8, device=DEV)
    xg, keep, info = x0.to(DEV).clone(), [], {}
    if general:
        hip.qk_attn(xg, rowmap, B_, Tq, N1, nH, plin, npj.p, nq.p, nk_.p, ng.p, keep_ws=keep, narrow=True, **kw)
    else:
        hip.qk_attn(xg, rowmap, B_, Tq, N1, nH, plin, npj.p, nq.p, nk_.p, ng.p, keep_ws=keep, x_src=zsrc, emit=(ws_mlp, ne.p), info=info, **kw)
    torch.cuda.synchronize()
    assert general or info.get("emitted") is True, "the wide-stage kernels were not taken"
    ws = keep[0].cpu()
    pad = lambda n: (n + 255) // 256 * 256
    e = ws[:M * Cc].view(Tq, rows, Cc)
    qkb = ws[pad(M * Cc):][:M * 2 * Cc]
    xs = ws[pad(M * Cc) + pad(M * 2 * Cc):][:M * Cc].view(Tq, rows, Cc)
    if general:                        # (the general kernels keep no copy of the slice spikes: the neuron kernel again, as the engine's tape does)
        xs_g = torch.empty((Tq, rows, Cc), dtype=torch.uint8, device=DEV)
        hip.neuron_fwd(x0.to(DEV), xs_g, Tq, 1, rows * Cc, 0, 0, 0, rows * Cc, npj.p, rowmap=rowmap, rowlen=Cc)
        xs = xs_g.cpu()
    if stacked:
        qs, ks = qkb.view(Tq, rows, 2 * Cc)[..., :Cc], qkb.view(Tq, rows, 2 * Cc)[..., Cc:]
    else:
        qs, ks = qkb[:M * Cc].view(Tq, rows, Cc), qkb[M * Cc:].view(Tq, rows, Cc)
    rm = rowmap.cpu().long()
    # (a) slice spikes: SN_proj over the T' steps of x gathered through the slice map (padding reads as 0)
    xg0 = torch.zeros((M, Cc))
    ok = rm >= 0
    xg0[ok] = x0.reshape(x_rows, Cc)[rm[ok]]
    xs_ref = npj.ref(xg0.view(Tq, rows, Cc))
    assert torch.equal(xs.float(), xs_ref), "slice spikes differ from the oracle"
    assert 0.03 < xs_ref.mean() < 0.97
    # (b) q | k on the kernel's own slice spikes
    xsd = xs.double().view(M, Cc)
    hq = ((xsd @ Wqe.t()) * aq.double() + bq.double()).view(Tq, rows, Cc)
    hk = ((xsd @ Wke.t()) * ak.double() + bk.double()).view(Tq, rows, Cc)
    hk = (hk.view(Tq, B_, N1, Cc) + pe.double().view(Tq, 1, N1, Cc)).view(Tq, rows, Cc)
    for h, got, what, nn in ((hq, qs, "q", nq), (hk, ks, "k", nk_)):
        rep = nn.check(h.float().contiguous(), got.float().contiguous())
        assert rep["unexplained"] == 0, (what, rep)
        assert rep["flips"] <= 2e-4 * got.numel(), (what, rep)
        assert 0.03 < got.float().mean() < 0.97, what
    # (c) token gate: exact on the kernel's own q and k
    a = qs.float().view(Tq, rows, nH, 32).sum(-1)
    gate = ng.ref(a.contiguous())
    assert name != "psn" or 0.03 < gate.mean() < 0.99
    e_ref = ks.float().view(Tq, rows, nH, 32) * gate.unsqueeze(-1)
    assert torch.equal(e.float().view(Tq, rows, nH, 32), e_ref), "gated spikes differ"
    assert 0.01 < e_ref.mean() < 0.9
    # (d) projection through the head scramble + BN + scatter + shortcut, on the kernel's own E
    Z = e.reshape(B_, nH, Tq, N1, 32).permute(2, 0, 3, 1, 4).reshape(M, Cc).double()
    Y = ((Z @ (_weff16(plin.Wp) if general else _weff(plin.digits)).t()) + biasp.double()) * ap.double() + bp.double()
    ref = x0.reshape(x_rows, Cc).double().clone()
    ref[rm[ok]] += Y[ok]
    err = (xg.cpu().reshape(x_rows, Cc).double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item(), err
    if general:
        return
    # (e) the emitted first neuron of the MLP: SN over D of the kernel's own updated x, bit-exact
    s1 = ws_mlp.cpu().view(B, D, H * W, Cc)
    s1_ref = ne.ref(xg.cpu().view(B, D, H * W, Cc).permute(1, 0, 2, 3).contiguous()).permute(1, 0, 2, 3)
    assert torch.equal(s1.float(), s1_ref), "emitted SN1 spikes differ from the oracle"
    # (f) the general kernels of the same entry point
    xb, kb = x0.to(DEV).clone(), []
    hip.qk_attn(xb, rowmap, B_, Tq, N1, nH, plin, npj.p, nq.p, nk_.p, ng.p, keep_ws=kb, narrow=True, **kw)
    torch.cuda.synchronize()
    eb = kb[0].cpu()[:M * Cc].view(Tq, rows, Cc)
    assert (eb != e).float().mean().item() <= 2e-4
    rows_same = ~(eb != e).any(-1).any(0)                                   # (rows,): tokens whose gated spikes agree at both steps
    assert rows_same.float().mean().item() > 0.9
    # the production form (no tape, nothing emitted) must give the same x bit for bit
    xh = x0.to(DEV).clone()
    hip.qk_attn(xh, rowmap, B_, Tq, N1, nH, plin, npj.p, nq.p, nk_.p, ng.p, x_src=zsrc, **kw)
    torch.cuda.synchronize()
    assert torch.equal(xh, xg), "the tape changes the result"



def test_wide_block_through_the_engine_matches_the_general_kernels():
    """One MS block at the stage-2 shape through `MSFlowEngine.swin_block`-style calls: the wide path with the MLP's first spikes
    handed over by the projection equals the wide path with its own SN1 launch bit for bit, and the general kernels up to
    near-threshold spike decisions."""
    B, D, H, W, Cc, window, shift = 1, 10, 18, 24, 384, (2, 9, 9), (1, 4, 4)
    nH, Tq, N1, x0, Wq, Wk, Wp, aq, bq, ak, bk, ap, bp, biasp, pe = _attn_case(B, D, H, W, Cc, window, shift, seed=40)
    Ch = 4 * Cc
    p = _N("lif", D).p
    plin = _L(Wp, ap, bp, biasp)
    wcat = torch.cat([Wq, Wk], 0).to(DEV).contiguous()
    qk = {"Wp": hip.split_weight(wcat, 2), "digits": hip.split_weight_i8x3(wcat), "alpha": torch.cat([aq, ak]).to(DEV),
          "beta": torch.cat([bq, bk]).to(DEV), "add": torch.cat([torch.zeros_like(pe), pe], -1).to(DEV).contiguous()}
    fc1 = _L(rnd((Ch, Cc), 861, -0.15, 0.15), rnd((Ch,), 863, 0.5, 1.5), rnd((Ch,), 864, -0.2, 0.2))
    fc2 = _L(rnd((Cc, Ch), 862, -0.05, 0.05), rnd((Cc,), 865, 0.5, 1.5), rnd((Cc,), 866, -0.2, 0.2))
    rowmap, B_ = hip.window_slice_map(B, D, H, W, window, shift, DEV)
    zsrc = hip.window_zsrc_map(rowmap, B_, Tq, N1, nH, B * D * H * W)

    def run(mode):
        x = x0.to(DEV).clone()
        if mode == "handover":
            ws, info = hip.ms_mlp_workspace(x, Ch), {}
            hip.qk_attn(x, rowmap, B_, Tq, N1, nH, plin, p, p, p, p, qk=qk, x_src=zsrc, emit=(ws, p), info=info)
            assert info.get("emitted") is True
            hip.ms_mlp(x, fc1, fc2, p, p, ws=ws, s1_ready=True)
        else:
            hip.qk_attn(x, rowmap, B_, Tq, N1, nH, plin, p, p, p, p, qk=qk, x_src=zsrc if mode == "wide" else None, narrow=mode == "narrow")
            hip.ms_mlp(x, fc1, fc2, p, p, narrow=mode == "narrow")
        torch.cuda.synchronize()
        return x
    xa, xb, xc = run("handover"), run("wide"), run("narrow")
    assert torch.equal(xa, xb) and not torch.equal(xa, x0.to(DEV))
    d = (xa - xc).abs().view(-1, Cc)
    close = (d.max(dim=1).values <= 2e-5 * xc.abs().max()).float().mean().item()
    assert close > 0.85, close                                              # rows untouched by a flipped spike agree to rounding


# ------------------------------------------------------------------------- fc2's emission of the next neuron + patch merging
@pytest.mark.parametrize("name", ALL)
@pytest.mark.parametrize("B,D,H,W,Cc", [(1, 10, 15, 20, 384), (1, 10, 8, 10, 768), (2, 10, 5, 7, 384), (1, 20, 6, 5, 384),
                                        (1, 10, 15, 20, 96), (2, 10, 5, 7, 192), (1, 20, 6, 5, 96)])
def test_wide_mlp_emits_the_next_layers_first_spikes(B, D, H, W, Cc, name):
    """SdfMsMlpDesc.emit_next: SN_next(x after the update), bit-equal to the oracle neuron on the kernel's OWN updated x (what the patch
    merging - reference Spiking_swin_transformer3D.py:970 - or the bottleneck's MS_ResBlock.sn1 - Spiking_modules.py:922 - computes
    first), and x itself unchanged by asking for it."""
    if name not in ("lif", "psn") and (D != 10 or B != 1):
        pytest.skip("the other neuron classes are covered on the shipped T = 10 shapes")
    Ch = 4 * Cc
    x0 = rnd((B, D, H, W, Cc), 900, -0.5, 1.0)
    fc1 = _L(rnd((Ch, Cc), 901, -0.15, 0.15), rnd((Ch,), 903, 0.5, 1.5), rnd((Ch,), 904, -0.2, 0.2))
    fc2 = _L(rnd((Cc, Ch), 902, -0.05, 0.05), rnd((Cc,), 905, 0.5, 1.5), rnd((Cc,), 906, -0.2, 0.2))
    p, nn = _N("lif", D).p, _N(name, D, v_th=0.25, seed=21, bias=-0.2)
    pn = nn.p
    buf = torch.full((B, D, H, W, Cc), 7, dtype=torch.uint8, device=DEV)
    xg = hip.ms_mlp(x0.to(DEV).clone(), fc1, fc2, p, p, emit_next=(buf, pn))
    xb = hip.ms_mlp(x0.to(DEV).clone(), fc1, fc2, p, p)
    torch.cuda.synchronize()
    assert torch.equal(xg, xb)
    ref = nn.ref(xg.cpu().permute(1, 0, 2, 3, 4).contiguous()).permute(1, 0, 2, 3, 4)
    assert torch.equal(buf.cpu().float(), ref), "emitted spikes differ from the oracle neuron on the updated x"
    assert 0.02 < ref.mean() < 0.98


def _merge_ref(sp, We, alpha, beta):
    """fp64 BN(cat_2x2(S) W^T) in the reference's order (Spiking_swin_transformer3D.py:960-972): zero padding to even sizes, the four
    strided slices x0..x3 = (0,0), (1,0), (0,1), (1,1), concatenation along channels."""
    B, D, H, W, Cc = sp.shape
    s = torch.nn.functional.pad(sp.double(), (0, 0, 0, W % 2, 0, H % 2))
    cat = torch.cat([s[:, :, 0::2, 0::2], s[:, :, 1::2, 0::2], s[:, :, 0::2, 1::2], s[:, :, 1::2, 1::2]], -1)
    return (cat @ We.t()) * alpha.double() + beta.double()


@pytest.mark.parametrize("B,D,H,W,Cc", [(1, 10, 15, 20, 384), (1, 10, 30, 40, 256), (2, 10, 7, 9, 384), (1, 20, 6, 5, 128),
                                        (1, 10, 1, 3, 128), (3, 20, 4, 4, 256), (1, 10, 36, 48, 192), (2, 10, 5, 7, 64), (1, 20, 3, 4, 320),
                                        # narrow stages (ms_res.hip: the quadrant of every 16-byte piece decoded per lane): the first merge
                                        # of the shipped model, odd sizes, T = 20, C = 160
                                        (1, 10, 72, 96, 96), (1, 10, 15, 21, 96), (1, 20, 6, 5, 96), (2, 10, 7, 9, 160)])
def test_wide_patch_merge_against_the_oracle(B, D, H, W, Cc):
    """sdf_ms_patch_merge_fwd on given spikes: exact integer sums -> fp64 reference to 1e-6 of the range, odd sizes included (15 x 20 is
    the shipped stage-2 map), and against the gather-map + spike GEMM path it replaces."""
    N = 2 * Cc
    sp = (rnd((B, D, H, W, Cc), 910, 0.0, 1.0) < 0.3).to(torch.uint8)
    Wr = rnd((N, 4 * Cc), 911, -0.08, 0.08)
    lin = _L(Wr, rnd((N,), 912, 0.5, 1.5), rnd((N,), 913, -0.2, 0.2))
    out = hip.ms_patch_merge(sp.to(DEV), lin)
    assert out is not None, "the wide merge refused a shape it is built for"
    torch.cuda.synchronize()
    assert tuple(out.shape) == (B, D, (H + 1) // 2, (W + 1) // 2, N)
    ref = _merge_ref(sp, _weff(lin.digits), lin.alpha.cpu(), lin.beta.cpu())
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 1e-6 * ref.abs().max().item(), err
    # the general path: neuron output gathered through merge_row_map, then the spike GEMM
    from sdformerflow_amd.STSwinNet_SNN.Spiking_swin_transformer3D import merge_row_map
    src, H2, W2 = merge_row_map(B, D, H, W)
    idx = torch.from_numpy(src.reshape(-1).astype("int64"))
    flat = torch.cat([sp.view(-1, Cc), sp.new_zeros(1, Cc)], 0)[idx].view(D * B * H2 * W2, 4 * Cc).to(DEV)        # (T, B, H2, W2, 4C)
    o2 = torch.empty((D * B * H2 * W2, N), dtype=torch.float32, device=DEV)
    hip.spike_gemm(flat, lin.Wp, o2, o2.shape[0], N, 4 * Cc, alpha=lin.alpha, beta=lin.beta)
    o2 = o2.view(D, B, H2, W2, N).permute(1, 0, 2, 3, 4)
    assert (out - o2).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_wide_patch_merge_refuses_what_it_is_not_built_for():
    lin = _L(rnd((160, 320), 920, -0.1, 0.1), rnd((160,), 921, 0.5, 1.5), rnd((160,), 922, -0.2, 0.2))
    sp = torch.zeros((1, 10, 8, 8, 80), dtype=torch.uint8, device=DEV)                  # C % 32 != 0
    assert hip.ms_patch_merge(sp, lin) is None
    lin2 = _L(rnd((256, 512), 923, -0.1, 0.1), rnd((256,), 924, 0.5, 1.5), rnd((256,), 925, -0.2, 0.2))
    assert hip.ms_patch_merge(torch.zeros((1, 4, 8, 8, 128), dtype=torch.uint8, device=DEV), lin2) is None    # D not in {10, 20}


@pytest.mark.parametrize("B,D,H,W,Cc", [(1, 10, 18, 24, 384), (1, 10, 9, 12, 768), (2, 20, 5, 3, 384), (3, 10, 4, 6, 192)])
def test_fc2_on_the_small_m_kernel_equals_the_wide_main_loop(B, D, H, W, Cc, monkeypatch):
    """`SdfMsMlpDesc.fc2_tiled`: fc2 on csrc/ms_smallm.hip (K split over the waves of a workgroup) against the wide main loop - the same
    integer sums and the same fp32 epilogue: x and the emitted next-layer spikes bit-equal."""
    Ch = 4 * Cc
    x0 = rnd((B, D, H, W, Cc), 930, -0.5, 1.0)
    fc1 = _L(rnd((Ch, Cc), 931, -0.15, 0.15), rnd((Ch,), 933, 0.5, 1.5), rnd((Ch,), 934, -0.2, 0.2))
    fc2 = _L(rnd((Cc, Ch), 932, -0.05, 0.05), rnd((Cc,), 935, 0.5, 1.5), rnd((Cc,), 936, -0.2, 0.2))
    p, pn = _N("lif", D).p, _N("lif", D, v_th=0.25).p

    def run():
        buf = torch.full((B, D, H, W, Cc), 7, dtype=torch.uint8, device=DEV)
        x = hip.ms_mlp(x0.to(DEV).clone(), fc1, fc2, p, p, emit_next=(buf, pn))
        y = hip.ms_mlp(x0.to(DEV).clone(), fc1, fc2, p, p)
        torch.cuda.synchronize()
        return x, buf, y
    monkeypatch.setenv("SDF_SMALLM_FC2", "2")                          # (at any size: the dispatcher takes it up to 512 tiles)
    xs, bs, ys = run()
    monkeypatch.setenv("SDF_SMALLM_FC2", "0")
    xw, bw, yw = run()
    assert torch.equal(xs, xw) and torch.equal(bs, bw) and torch.equal(ys, yw) and torch.equal(xs, ys)
    assert not torch.equal(xs, x0.to(DEV))


# ----------------------------------------------------------- narrow-stage kernels (ms_res.hip) against the wide-stage kernels (ms_wide.hip)
@pytest.mark.parametrize("name", ["lif", "psn"])
def test_resident_kernels_equal_the_wide_kernels_bit_for_bit(name, monkeypatch):
    """C = 192 (swin stage 1) is served by both kernel families: the weight-resident row-loop kernels (default) and the wide-stage K-ring
    kernels (SDF_RES=0).  Same exact integer sums, same fp32 epilogue expressions: the block's x, the gated spikes E, the SN1 spikes the
    projection emits, the hidden spikes, the emitted next-layer spikes and the patch merging agree bit for bit."""
    B, D, H, W, Cc, window, shift = 1, 10, 36, 48, 192, (2, 9, 9), (1, 4, 4)
    nH, Tq, N1, x0, Wq, Wk, Wp, aq, bq, ak, bk, ap, bp, biasp, pe = _attn_case(B, D, H, W, Cc, window, shift, seed=60)
    Ch = 4 * Cc
    npj, nq, nk_, ng = (_N(name, Tq, seed=31 + i, **({"gain": 0.12, "bias": -1.2} if i == 3 else {})) for i in range(4))
    n1, n2, nn = _N(name, D, seed=41), _N(name, D, seed=42), _N(name, D, v_th=0.25, seed=43, bias=-0.2)
    plin, qlin, klin = _L(Wp, ap, bp, biasp), _L(Wq, aq, bq), _L(Wk, ak, bk)
    fc1 = _L(rnd((Ch, Cc), 961, -0.15, 0.15), rnd((Ch,), 963, 0.5, 1.5), rnd((Ch,), 964, -0.2, 0.2))
    fc2 = _L(rnd((Cc, Ch), 962, -0.05, 0.05), rnd((Cc,), 965, 0.5, 1.5), rnd((Cc,), 966, -0.2, 0.2))
    mlin = _L(rnd((2 * Cc, 4 * Cc), 967, -0.08, 0.08), rnd((2 * Cc,), 968, 0.5, 1.5), rnd((2 * Cc,), 969, -0.2, 0.2))
    rowmap, B_ = hip.window_slice_map(B, D, H, W, window, shift, DEV)
    zsrc = hip.window_zsrc_map(rowmap, B_, Tq, N1, nH, B * D * H * W)

    def run():
        x = x0.to(DEV).clone()
        ws, info, keep = hip.ms_mlp_workspace(x, Ch), {}, []
        hip.qk_attn(x, rowmap, B_, Tq, N1, nH, plin, npj.p, nq.p, nk_.p, ng.p, q_lin=qlin, k_lin=klin, pe=pe.to(DEV).contiguous(), x_src=zsrc,
                    emit=(ws, n1.p), info=info, keep_ws=keep)
        assert info.get("emitted") is True
        xa = x.clone()
        s1 = ws[:x.numel()].clone()                               # (row-major: the tape form)
        kb = []
        hip.ms_mlp(x, fc1, fc2, n1.p, n2.p, keep_ws=kb)
        buf = torch.full((B, D, H, W, Cc), 7, dtype=torch.uint8, device=DEV)
        x2 = hip.ms_mlp(xa.clone(), fc1, fc2, n1.p, n2.p, emit_next=(buf, nn.p))
        mg = hip.ms_patch_merge(buf, mlin)
        torch.cuda.synchronize()
        M, pad = Tq * B_ * N1, lambda n: (n + 255) // 256 * 256
        wsa, wsm, ntok = keep[0], kb[0], B * D * H * W             # (only the regions the calls define: the paddings between them are not written)
        return (xa, wsa[:M * Cc].clone(), wsa[pad(M * Cc):][:2 * M * Cc].clone(), wsa[pad(M * Cc) + pad(2 * M * Cc):][:M * Cc].clone(), s1, x,
                wsm[:ntok * Cc].clone(), wsm[pad(ntok * Cc):][:ntok * Ch].clone(), x2, buf, mg)
    a = run()
    monkeypatch.setenv("SDF_RES", "0")
    b = run()
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), f"output {i} differs between the resident and the wide kernels"
    assert not torch.equal(a[0], x0.to(DEV)) and 0.02 < a[9].float().mean() < 0.98
