"""`sdf_spike_conv2d_multi_fwd`: the four output-parity classes of MS_SpikingTransposeDecoderLayer's ConvTranspose2d(k 3, s 2, p 1, op 1)
(reference Spiking_modules.py:461-474) as ONE launch of the streaming kernel - bit-equal to the four single launches (same kernel body,
same work items per class), and equal to torch's transposed convolution of the same spikes in fp64 to the two-plane weight precision."""
import pytest
import torch

from sdformerflow_amd import hip
from sdformerflow_amd.engine import deconv_classes
from sdformerflow_amd.synthetic import synth_uniform as rnd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("imgs,H,W,Cin,Cout", [(10, 72, 96, 196, 96), (10, 12, 16, 100, 96), (4, 9, 7, 64, 192)])
def test_parity_classes_in_one_launch(imgs, H, W, Cin, Cout, monkeypatch):
    cp = (Cin + 15) // 16 * 16
    s = torch.zeros((imgs, H, W, cp), dtype=torch.uint8)
    s[..., :Cin] = (rnd((imgs, H, W, Cin), 500) < 0.3).to(torch.uint8)
    w = rnd((Cin, Cout, 3, 3), 501, -0.08, 0.08)
    alpha, beta = rnd((Cout,), 502, 0.5, 1.5).to(DEV), rnd((Cout,), 503, -0.2, 0.2).to(DEV)
    classes = deconv_classes(w.to(DEV), imgs, H, W, cp, 2, DEV)
    sd = s.to(DEV)

    def run(multi):
        z = torch.full((imgs, 4 * H * W, Cout), float("nan"), device=DEV)
        if multi:
            hip.spike_conv2d_multi(sd, classes, imgs, H, W, cp, H, W, z, alpha=alpha, beta=beta)
        else:
            for c in classes:
                hip.spike_conv2d(sd, c["Wp"], imgs, H, W, cp, H, W, c["KH"], c["KW"], 1, c["dy"], c["dx"], out=z, alpha=alpha, beta=beta,
                                 out_rowmap=c["rowmap"])
        torch.cuda.synchronize()
        return z
    one, four = run(True), run(False)
    assert torch.equal(one, four) and not torch.isnan(one).any()
    monkeypatch.setenv("SDF_CONV_MULTI", "0")                   # the entry point's own one-by-one path
    assert torch.equal(run(True), four)
    ref = torch.nn.functional.conv_transpose2d(s[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), None, 2, 1, 1)
    ref = ref.permute(0, 2, 3, 1) * alpha.cpu().double() + beta.cpu().double()
    err = (one.cpu().view(imgs, 2 * H, 2 * W, Cout).double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item(), err


@pytest.mark.parametrize("imgs,T,H,W,Cin,Cout", [(10, 10, 72, 96, 196, 48), (10, 10, 21, 37, 196, 48), (10, 10, 12, 16, 100, 96), (20, 10, 9, 7, 64, 48), (20, 20, 5, 6, 196, 48),
                                                  (10, 10, 1, 1, 48, 8), (10, 10, 7, 1, 250, 24)])
def test_transposed_convolution_as_one_product(imgs, T, H, W, Cin, Cout, monkeypatch):
    """`sdf_spike_deconv3x3s2_fwd` (round 5): the same ConvTranspose2d(3, 2, 1, 1) + BatchNorm as ONE digit product - a row is an input
    pixel, its K the 2 x 2 input neighbourhood (zero beyond the image: the last row / column, and every neighbour of a 1 x 1 image), its
    4 Cout columns the 2 x 2 output block.  Within 1e-5 of fp64 on the output's magnitude (23-bit digits, exact integer sums) and of the
    parity-class launches; every output element written (NaN pre-fill)."""
    cp = (Cin + 15) // 16 * 16
    s = torch.zeros((imgs, H, W, cp), dtype=torch.uint8)
    s[..., :Cin] = (rnd((imgs, H, W, Cin), 510 + H) < 0.3).to(torch.uint8)
    w = rnd((Cin, Cout, 3, 3), 511, -0.08, 0.08)
    alpha, beta = rnd((Cout,), 512, 0.5, 1.5).to(DEV), rnd((Cout,), 513, -0.2, 0.2).to(DEV)
    assert hip.deconv2x2_applicable(imgs, T, H, W, cp, Cout)
    planes = hip.pack_deconv2x2_weight(w.to(DEV), cp)
    z = torch.full((imgs, 2 * H, 2 * W, Cout), float("nan"), device=DEV)
    hip.spike_deconv3x3s2(s.to(DEV), planes, imgs, T, H, W, cp, Cout, alpha=alpha, beta=beta, out=z)
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv_transpose2d(s[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), None, 2, 1, 1)
    ref = ref.permute(0, 2, 3, 1) * alpha.cpu().double() + beta.cpu().double()
    got = z.cpu().double()
    assert not torch.isnan(got).any()
    assert (got - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # bit-equal to itself run to run (exact integer sums), and - on weights both formats hold exactly - to the parity-class launches
    z2 = torch.empty_like(z)
    hip.spike_deconv3x3s2(s.to(DEV), planes, imgs, T, H, W, cp, Cout, alpha=alpha, beta=beta, out=z2)
    assert torch.equal(z, z2)
    # the halo-tile kernel (208 padded channels, >= 4 096 pixels) and the row-loop kernel's form of the product: the same exact sums
    monkeypatch.setenv("SDF_DECONV_WRES", "0")
    z3 = torch.full_like(z, float("nan"))
    hip.spike_deconv3x3s2(s.to(DEV), planes, imgs, T, H, W, cp, Cout, alpha=alpha, beta=beta, out=z3)
    monkeypatch.delenv("SDF_DECONV_WRES")
    assert torch.equal(z, z3)
    if Cout % 96 == 0:
        wq = (w * 1024).round() / 1024
        zq = hip.spike_deconv3x3s2(s.to(DEV), hip.pack_deconv2x2_weight(wq.to(DEV), cp), imgs, T, H, W, cp, Cout)
        zc = torch.full((imgs, 4 * H * W, Cout), float("nan"), device=DEV)
        hip.spike_conv2d_multi(s.to(DEV), deconv_classes(wq.to(DEV), imgs, H, W, cp, 2, DEV), imgs, H, W, cp, H, W, zc)
        assert torch.equal(zq.view(imgs, 4 * H * W, Cout), zc)
