import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from sdformerflow_amd import hip
g = torch.Generator().manual_seed(0)
bad = 0
for (imgs, H, W) in [(1, 1, 1), (1, 2, 3), (3, 5, 9), (2, 4, 8), (1, 7, 17), (5, 3, 31)]:
    for Cin, Cout in [(16, 32), (96, 64)]:
        x = torch.randn(imgs, Cin, H, W, generator=g).cuda(); w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 10).cuda()
        y = hip.unpack_planes(hip.dense_conv3x3(hip.pack_planes(x), hip.pack_dense_conv_weight(w)), Cout)
        ref = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
        e = (y.double().cpu() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
        if e > 1e-5: bad += 1
        print(imgs, H, W, Cin, Cout, f"{e:.1e}")
for (M, K, N) in [(1, 32, 96), (7, 64, 96), (129, 96, 192), (257, 32, 96)]:
    a = torch.randn(M, K, generator=g).cuda(); w = torch.randn(N, K, generator=g).cuda() / 8
    y = hip.dense_linear(a, hip.pack_dense_linear_weight(w))
    ref = a.double().cpu() @ w.double().cpu().T
    e = (y.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    if e > 1e-5: bad += 1
    print("linear", M, K, N, f"{e:.1e}")
for rows, C in [(1, 4), (3, 8), (1, 2048), (2, 1028)]:
    x = torch.randn(rows, C, generator=g).cuda(); wv = torch.rand(C, generator=g).cuda(); b = torch.randn(C, generator=g).cuda()
    y = hip.layer_norm(x, wv, b, 1e-5); ref = F.layer_norm(x.double().cpu(), (C,), wv.double().cpu(), b.double().cpu(), 1e-5)
    e = (y.double().cpu() - ref).abs().max().item()
    if e > 1e-5: bad += 1
    print("ln", rows, C, f"{e:.1e}")
print("BAD", bad)
