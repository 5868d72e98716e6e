#!/usr/bin/env python3
"""One MS swin block at a wide-stage shape through the C ABI (attention with the projection's hand-over + MLP), a few times - for
rocprofv3 passes and tools/stamp_wide.sh.  usage: wide_one.py [B D H W C] [narrow] | wide_one.py conv [B T H W Cin Cout]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd


class L:
    def __init__(self, W, alpha, beta, bias=None):
        self.N, self.K = W.shape
        self.Wp = hip.split_weight(W.to("cuda:0").contiguous(), 2)
        self.digits = hip.split_weight_i8x3(W.to("cuda:0").contiguous())
        if self.K % 64 == 0 and self.N % 16 == 0:
            self.digits_tiled = hip.tile_weight_i8x3(self.digits)
        self.alpha, self.beta = alpha.to("cuda:0"), beta.to("cuda:0")
        self.bias = None if bias is None else bias.to("cuda:0")


def block(B, D, H, W, Cc, window=(2, 9, 9), shift=(1, 4, 4)):
    dev = "cuda:0"
    nH, Tq, N1, Ch = Cc // 32, window[0], window[1] * window[2], 4 * Cc
    x = rnd((B, D, H, W, Cc), 1, -0.5, 1.0).to(dev)
    Wq, Wk, Wp = (rnd((Cc, Cc), 2 + i, -0.12, 0.12) for i in range(3))
    pe = rnd((Tq * N1, Cc), 9, -0.2, 0.2)
    wcat = torch.cat([Wq, Wk], 0).to(dev).contiguous()
    qk = {"Wp": hip.split_weight(wcat, 2), "digits": hip.split_weight_i8x3(wcat), "alpha": rnd((2 * Cc,), 5, 0.5, 1.5).to(dev),
          "beta": rnd((2 * Cc,), 6, -0.1, 0.3).to(dev), "add": torch.cat([torch.zeros_like(pe), pe], -1).to(dev).contiguous()}
    plin = L(Wp, rnd((Cc,), 7, 0.5, 1.5), rnd((Cc,), 8, -0.2, 0.2), rnd((Cc,), 10, -0.1, 0.1))
    fc1 = L(rnd((Ch, Cc), 11, -0.15, 0.15), rnd((Ch,), 12, 0.5, 1.5), rnd((Ch,), 13, -0.2, 0.2))
    fc2 = L(rnd((Cc, Ch), 14, -0.05, 0.05), rnd((Cc,), 15, 0.5, 1.5), rnd((Cc,), 16, -0.2, 0.2))
    p = hip.NeuronParams("lif", 2.0, 0.1, None)
    rowmap, B_ = hip.window_slice_map(B, D, H, W, window, shift, dev)
    zsrc = hip.window_zsrc_map(rowmap, B_, Tq, N1, nH, B * D * H * W)

    def run(narrow=False):
        ws, info = hip.ms_mlp_workspace(x, Ch), {}
        hip.qk_attn(x, rowmap, B_, Tq, N1, nH, plin, p, p, p, p, qk=qk, x_src=None if narrow else zsrc, emit=None if narrow else (ws, p),
                    info=info, narrow=narrow)
        hip.ms_mlp(x, fc1, fc2, p, p, ws=ws, s1_ready=bool(info.get("emitted")), narrow=narrow)
    return run


def conv(B, T, H, W, Cin, Cout):
    """The U-Net bottleneck's 3x3 spike convolution with the fused BN + LIF epilogue (small M, digit planes)."""
    dev = "cuda:0"
    imgs, hw = B * T, H * W
    x = (rnd((imgs, H, W, Cin), 21) < -0.4).to(torch.uint8).to(dev)
    dg = hip.pack_conv_weight_i8x3(rnd((Cout, Cin, 3, 3), 22, -0.05, 0.05).to(dev), tiled=os.environ.get("SDF_WIDE_CONV", "") != "1" and os.environ.get("ROWMAJOR", "") != "1")
    al, be = rnd((Cout,), 23, 0.5, 1.5).to(dev), rnd((Cout,), 24, -0.1, 0.3).to(dev)
    sp = torch.empty((imgs * hw, Cout), dtype=torch.uint8, device=dev)
    p = hip.NeuronParams("lif", 2.0, 0.1, None)

    def run():
        hip.spike_conv2d(x, dg, imgs, H, W, Cin, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out_spike=sp, alpha=al, beta=be, sn=p, sn_T=T,
                         pos=(B * hw, hw, T * hw, hw))
    return run


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "conv":          # wide_one.py conv [B T H W Cin Cout]: the small-M convolution alone
        a = [int(v) for v in sys.argv[2:8]] if len(sys.argv) >= 8 else [1, 10, 9, 12, 768, 768]
        run = conv(*a)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        sys.exit(0)
    a = [int(v) for v in sys.argv[1:6]] if len(sys.argv) >= 6 else [1, 10, 18, 24, 384]
    run = block(*a)
    for _ in range(10):
        run(narrow="narrow" in sys.argv)
    torch.cuda.synchronize()
