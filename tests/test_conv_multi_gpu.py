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
