import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sdformerflow_amd import hip
from sdformerflow_amd.synthetic import synth_uniform as rnd
from oracle import sdformer_oracle as O
DEV = "cuda:0"
nH, N, nW, B = 3, 162, 4, 2
Cc = nH * 32
qkv = rnd((B * nW, N, 3 * Cc), 60, -1.0, 1.0)
ls = torch.exp(torch.clamp(rnd((nH, 1, 1), 61, 1.5, 5.0), max=float(np.log(100.0))))
bias = 16 * torch.sigmoid(rnd((nH, N, N), 62, -2.0, 2.0))
mask = O.compute_mask(2, 18, 18, (2, 9, 9), (1, 4, 4))
print("ls", ls.reshape(-1))
for m in (mask, None):
    ref, _ = O.ann_attention_core(qkv, ls, bias, m, nH)
    for env in ("", "1"):
        os.environ["SDF_ATTN_F32"] = env
        got = hip.win_attn_ann(qkv.to(DEV), ls.reshape(-1).contiguous().to(DEV), bias.to(DEV), m.to(DEV) if m is not None else None, nH).cpu()
        e = (got - ref).abs()
        idx = np.unravel_index(int(e.argmax()), e.shape)
        print("mask" if m is not None else "nomask", "f32" if env else "f16", "max err", e.max().item(), "at", idx, "mean", e.mean().item(), "nan", torch.isnan(got).sum().item())
os.environ["SDF_ATTN_F32"] = ""
ref, _ = O.ann_attention_core(qkv, ls, bias, mask, nH)
got = hip.win_attn_ann(qkv.to(DEV), ls.reshape(-1).contiguous().to(DEV), bias.to(DEV), mask.to(DEV), nH).cpu()
e = (got - ref).abs().view(B * nW, N, nH, 32)
print("err per window", e.amax((1, 2, 3)))
print("err per head", e.amax((0, 1, 3)))
print("err per qtile", e.view(B * nW, N, -1).amax((0, 2)).view(-1)[:160].view(10, 16).amax(1))
print("masked fraction per window", (mask != 0).float().mean((1, 2)))
z = torch.zeros_like(mask)
ref0, _ = O.ann_attention_core(qkv, ls, bias, None, nH)
got0 = hip.win_attn_ann(qkv.to(DEV), ls.reshape(-1).contiguous().to(DEV), bias.to(DEV), z.to(DEV), nH).cpu()
print("zero mask err", (got0 - ref0).abs().max().item())
e0 = (got0 - ref0).abs().view(B * nW, N, nH, 32)
torch.set_printoptions(linewidth=250, precision=1, sci_mode=False)
print("zero-mask err per token, window 0:", (e0[0].amax((1, 2)) > 1e-4).int().tolist())
print("zero-mask err per token, window 5:", (e0[5].amax((1, 2)) > 1e-4).int().tolist())
print("per dim (window 0, token 0):", e0[0, 0])
print("got", got0.view(B * nW, N, nH, 32)[0, 0, 0, :8], "ref", ref0.view(B * nW, N, nH, 32)[0, 0, 0, :8])
