"""GPU parity of the one-launch flow-prediction head (csrc/pred_head.hip) through the C ABI `sdf_pred_head_fwd`, against the
reference's expression SN -> conv1x1 + bias -> sum over T -> nearest upsampling (Spiking_modules.py:605-640,
Spiking_STSwinNet.py:289-303) and the next decoder level's neuron on cat(pred, z) (Spiking_STSwinNet.py:168-172):
spikes bit-equal to the C oracle neuron, predictions / flows to 1e-5 of their range against fp64 on the kernel's own spikes."""
import pytest
import torch

from oracle import neuron_ref as R
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("kind", ["lif", "psn"])
@pytest.mark.parametrize("B,D,h,w,Cin,scale", [(1, 10, 9, 12, 96, 2), (2, 10, 5, 7, 192, 4), (1, 10, 3, 4, 384, 16), (1, 20, 6, 5, 96, 2),
                                                (2, 5, 4, 9, 192, 1)])
def test_pred_head_against_the_oracle(kind, B, D, h, w, Cin, scale):
    if kind == "psn" and D > 10:
        pytest.skip("PSN over T = 20 keeps the three-launch form")
    z = rnd((B, D, h, w, Cin), 700 + Cin + D, -0.5, 1.0)
    wgt, bias = rnd((2, Cin), 701, -0.3, 0.3), rnd((2,), 702, -0.2, 0.2)
    Wn, bn = rnd((D, D), 703, -0.5, 0.5) + 0.5 * torch.eye(D), torch.full((D,), -0.1)
    Wm = rnd((D, D), 704, -0.4, 0.6) + 0.4 * torch.eye(D)
    sn = hip.NeuronParams(kind, 2.0, 0.1, None, Wn.to(DEV), bn.to(DEV))
    sn_next = hip.NeuronParams(kind, 2.0, 0.15 if kind == "lif" else 0.1, None, Wm.to(DEV), bn.to(DEV))     # a DIFFERENT neuron
    H, W = h * scale, w * scale
    C2, ld = 32, Cin + 32 + 4 + 12
    img = torch.full((B, D, h, w, ld), 7, dtype=torch.uint8, device=DEV)
    nxt = (img, sn_next, 0, Cin + C2, (Cin + C2 + 4, 12))
    pred, flow, sp = hip.pred_head(z.to(DEV), wgt.to(DEV), bias.to(DEV), sn, H, W, want_pred=True, nxt=nxt, keep=True)
    torch.cuda.synchronize()
    zt = z.permute(1, 0, 2, 3, 4).contiguous()
    ref_sp = R.neuron_ref(zt, kind, 2.0, 0.1, None, psn_w=Wn, psn_b=bn).permute(1, 0, 2, 3, 4)
    assert torch.equal(sp.cpu().float(), ref_sp), "SN_pred(z) differs from the oracle"
    assert 0.03 < ref_sp.mean() < 0.97
    p64 = ref_sp.double() @ wgt.double().t() + bias.double()                                    # (B,D,h,w,2)
    got = pred.cpu()
    assert torch.equal(got[..., 2:], torch.zeros_like(got[..., 2:]))
    assert (got[..., :2].double() - p64).abs().max() <= 1e-5 * p64.abs().max()
    f64 = torch.nn.functional.interpolate(p64.sum(1).permute(0, 3, 1, 2), scale_factor=(H / h, W / w))
    assert flow.shape == (B, 2, H, W) and (flow.cpu().double() - f64).abs().max() <= 1e-5 * f64.abs().max()
    # the next level's image: SN_next(z) in [0, Cin), SN_next(pred) in the 4-wide slice, zeros in the padding, the skip slice untouched
    im = img.cpu()
    v_th = 0.15 if kind == "lif" else 0.1
    ref_nz = R.neuron_ref(zt, kind, 2.0, v_th, None, psn_w=Wm, psn_b=bn).permute(1, 0, 2, 3, 4)
    assert torch.equal(im[..., :Cin].float(), ref_nz)
    ref_np = R.neuron_ref(got[..., :2].permute(1, 0, 2, 3, 4).contiguous(), kind, 2.0, v_th, None, psn_w=Wm, psn_b=bn).permute(1, 0, 2, 3, 4)
    assert torch.equal(im[..., Cin + C2:Cin + C2 + 2].float(), ref_np)
    assert int(im[..., Cin + C2 + 2:].sum()) == 0 and bool((im[..., Cin:Cin + C2] == 7).all())
    # the same neuron on both sides, and no optional outputs: same flow
    img2 = torch.zeros_like(img)
    _, flow2, _ = hip.pred_head(z.to(DEV), wgt.to(DEV), bias.to(DEV), sn, H, W, want_pred=False, nxt=(img2, sn, 0, Cin + C2, (Cin + C2 + 4, 12)))
    torch.cuda.synchronize()
    assert torch.equal(flow2, flow) and torch.equal(img2[..., :Cin], sp)


def test_pred_head_refuses_what_it_has_no_kernel_for():
    z = torch.zeros((1, 10, 4, 4, 64), device=DEV)
    sn = hip.NeuronParams("lif", 2.0, 0.1, None)
    with pytest.raises(hip.SdfError):
        hip.pred_head(z, torch.zeros((2, 64), device=DEV), None, sn, 8, 8)                  # Cin = 64
    z = torch.zeros((1, 10, 4, 4, 96), device=DEV)
    with pytest.raises(hip.SdfError):
        hip.pred_head(z, torch.zeros((2, 96), device=DEV), None, sn, 6, 8)                  # 6 / 4 is not a whole factor
