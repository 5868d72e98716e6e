#!/usr/bin/env python3
"""Ping-pong kernel / fp16-plane check on the GPU box: correctness vs fp64 and vs the barrier kernel, timings."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdformerflow_amd import hip
dev = "cuda:0"
torch.manual_seed(0)

def timeit(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def conv_case(imgs, H, W, Cin, Cout, stride=1, resid=True, fused=False, kind="lif"):
    x = (torch.rand((imgs, H, W, Cin), device=dev) < 0.3).to(torch.uint8)
    w = torch.randn((Cout, Cin, 3, 3), device=dev) * 0.05
    al, be = torch.rand(Cout, device=dev) + 0.5, torch.randn(Cout, device=dev) * 0.1
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    M = imgs * OH * OW
    res = torch.randn((M, Cout), device=dev) if resid else None
    # fp64 reference
    xr = x.permute(0, 3, 1, 2).double()
    ref = torch.nn.functional.conv2d(xr, w.double(), None, stride, 1).permute(0, 2, 3, 1).reshape(M, Cout)
    ref32 = torch.nn.functional.conv2d(xr.float(), w, None, stride, 1).permute(0, 2, 3, 1).reshape(M, Cout)
    print(f"conv imgs={imgs} {H}x{W} {Cin}->{Cout} s={stride} resid={resid} fused={fused} {kind if fused else ''}")
    print(f"   torch fp32 conv vs fp64: max abs err {float((ref32.double() - ref).abs().max()):.3e}  (mean |ref| {float(ref.abs().mean()):.3f})")
    outs = {}
    for name, nsplit, pp in (("ws3", 3, "0"), ("pp3", 3, "1"), ("pp2", 2, "1")):
        os.environ["SDF_CONV_PP"] = pp
        Wp = hip.pack_conv_weight(w, nsplit)
        if fused:
            n = OH * OW
            out = torch.zeros((M, Cout), dtype=torch.uint8, device=dev)
            if kind == "psn":
                g = torch.Generator(device="cpu").manual_seed(5)
                sn = hip.NeuronParams("psn", 2.0, 0.1, None, psn_w=(torch.randn((10, 10), generator=g) * 0.3).to(dev),
                                      psn_b=(torch.randn((10,), generator=g) * 0.1 - 0.1).to(dev))
            else:
                sn = hip.NeuronParams("lif", 2.0, 0.1, None)
            f = lambda: hip.spike_conv2d(x, Wp, imgs, H, W, Cin, OH, OW, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out_spike=out, alpha=al,
                                         beta=be, sn=sn, sn_T=10, pos=(n, n, 0, n))
        else:
            out = torch.zeros((M, Cout), device=dev)
            f = lambda: hip.spike_conv2d(x, Wp, imgs, H, W, Cin, OH, OW, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=out, alpha=al,
                                         beta=be, resid=res)
        us = timeit(f)
        torch.cuda.synchronize()
        outs[name] = out.clone()
        fl = 2.0 * M * Cout * 9 * Cin
        msg = f"   {name}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF algorithmic"
        if not fused:
            full = ref * al.double() + be.double() + (res.double() if resid else 0)
            msg += f"   max abs err vs fp64 {float((out.double() - full).abs().max()):.3e}"
        print(msg)
    if fused:
        a, b, c = outs["ws3"], outs["pp2"], outs["pp3"]
        print(f"   spikes pp3 vs ws3: mismatch {float((a != c).float().mean()):.3e} (same planes: must be 0);  pp2 vs ws3: {float((a != b).float().mean()):.3e}  rate {float(a.float().mean()):.3f}")
        if (a != c).any():
            idx = (a != c).nonzero()
            n = OH * OW
            t, pos, col = idx[:, 0] // n, idx[:, 0] % n, idx[:, 1]
            print("   mismatches:", len(idx), " t histogram", torch.bincount(t, minlength=10).tolist())
            print("   col%32 histogram", torch.bincount(col % 32, minlength=32).tolist())
            print("   (pos%24)//3 [=2*wave+half] histogram", torch.bincount((pos % 24) // 3, minlength=8).tolist(), " pos%3 [=pl]", torch.bincount(pos % 3, minlength=3).tolist())
            print("   ws3 value at mismatches (mean):", float(a[idx[:, 0], idx[:, 1]].float().mean()))
    else:
        print(f"   pp3 == ws3 bit-exact: {bool((outs['pp3'] == outs['ws3']).all())};  pp2 vs ws3 max abs {float((outs['pp2'] - outs['ws3']).abs().max()):.3e}")

def gemm_case(M, N, K, resid=True):
    A = (torch.rand((M, K), device=dev) < 0.3).to(torch.uint8)
    w = torch.randn((N, K), device=dev) * 0.05
    al, be = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev) * 0.1
    res = torch.randn((M, N), device=dev) if resid else None
    ref = (A.double() @ w.double().t()) * al.double() + be.double() + (res.double() if resid else 0)
    print(f"gemm M={M} N={N} K={K} resid={resid}")
    for name, nsplit, ws in (("base3", 3, "0"), ("base2", 2, "0"), ("ws3", 3, "1"), ("pp3", 3, "2"), ("pp2", 2, "2")):
        os.environ["SDF_GEMM_WS"] = ws
        Wp = hip.split_weight(w, nsplit)
        out = torch.zeros((M, N), device=dev)
        f = lambda: hip.spike_gemm(A, Wp, out, M, N, K, alpha=al, beta=be, resid=res)
        us = timeit(f)
        print(f"   {name}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF   max abs err vs fp64 {float((out.double() - ref).abs().max()):.3e}")
    os.environ["SDF_GEMM_WS"] = "0"

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which == "psn":
    conv_case(10, 72, 96, 96, 96, 1, False, fused=True, kind="psn")
    conv_case(10, 72, 96, 96, 96, 1, False, fused=True, kind="lif")
if which in ("all", "conv"):
    conv_case(10, 144, 192, 96, 96, 1, True)
    conv_case(10, 144, 192, 96, 96, 1, False)
    conv_case(10, 144, 192, 96, 96, 1, False, fused=True)
    conv_case(10, 37, 53, 96, 192, 2, True)
    conv_case(10, 18, 24, 384, 384, 1, True)
if which in ("all", "gemm"):
    gemm_case(69120, 96, 384)
    gemm_case(69120, 384, 96, False)
    gemm_case(17280, 192, 768)
    gemm_case(1000, 768, 3072)
