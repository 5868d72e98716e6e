"""Parameter tree of the spiking 3-D swin encoder (mirror of reference
models/STSwinNet_SNN/Spiking_swin_transformer3D.py: QK token-gating attention :605-717, MS MLP :164-181,
MS block :720-895, MS patch merging :952-974, stage :995-1129, backbone :1132-1292).

Execution is scheduled by `sdformerflow_amd.engine`: pad / roll / window_partition_v2 / window_reverse
become row maps inside the kernels, BN is folded into neuron prologues and GEMM epilogues.
"""
import numpy as np
import torch
import torch.nn as nn

from .Spiking_modules import (MS_PED_Spiking_PatchEmbed_Conv_sfn, SpikingNormLayer, Spiking_neuron,  # noqa: F401
                              _neuron_kwargs)


def get_window_size(x_size, window_size, shift_size=None):
    """Clamp the window (and zero the shift) on axes that are not larger than it (reference
    models/STSwinNet/swin_transformer3D_v2.py:68-81)."""
    ws = [min(w, s) for w, s in zip(window_size, x_size)]
    if shift_size is None:
        return tuple(ws)
    ss = [0 if s <= w else sh for s, w, sh in zip(x_size, window_size, shift_size)]
    return tuple(ws), tuple(ss)


def window_slice_map(B, D, H, W, ws, ss):
    """int32 (Wd*B_, Wh*Ww) table: source token of every window-slice token, -1 = zero padding.
    Equivalent of F.pad + torch.roll(-shift) + window_partition_v2 + raw .view(Wd, B_, ...) of the
    reference (Spiking_swin_transformer3D.py:789-804, :100-113); used forwards as the gather map of the
    attention input and backwards as the scatter map of window_reverse + roll(+shift) + crop (:810-820)."""
    Wd, Wh, Ww = ws
    Dp, Hp, Wp = -(-D // Wd) * Wd, -(-H // Wh) * Wh, -(-W // Ww) * Ww
    nD, nHb, nWb = Dp // Wd, Hp // Wh, Wp // Ww
    B_ = B * nD * nHb * nWb
    j = np.arange(B_ * Wd, dtype=np.int64)
    wd, win = j % Wd, j // Wd
    wb, hb = win % nWb, (win // nWb) % nHb
    db, b = (win // (nWb * nHb)) % nD, win // (nWb * nHb * nD)
    tok = np.arange(Wh * Ww, dtype=np.int64)
    th, tw = tok // Ww, tok % Ww
    d = ((db * Wd + wd + ss[0]) % Dp)[:, None]
    h = (hb[:, None] * Wh + th[None, :] + ss[1]) % Hp
    w = (wb[:, None] * Ww + tw[None, :] + ss[2]) % Wp
    src = ((b[:, None] * D + d) * H + h) * W + w
    src = np.where((d < D) & (h < H) & (w < W), src, -1)
    return src.astype(np.int32), B_


def merge_row_map(B, D, H, W):
    """int32 (D, B*H2*W2*4) gather map of MS_SpikingPatchMerging's 2x2 concat in (T,B,H/2,W/2,4C) order
    (reference :965-970): quadrant q of the 4C axis is (dh,dw) = (q%2, q//2); odd sizes read zero (-1)."""
    H2, W2 = (H + 1) // 2, (W + 1) // 2
    t, b, h2, w2, q = np.meshgrid(np.arange(D), np.arange(B), np.arange(H2), np.arange(W2), np.arange(4), indexing="ij")
    h, w = 2 * h2 + q % 2, 2 * w2 + q // 2
    src = ((b * D + t) * H + h) * W + w
    return np.where((h < H) & (w < W), src, -1).reshape(D, -1).astype(np.int32), H2, W2


class Spiking_QK_WindowAttention3D(nn.Module):
    """Token-gating spiking attention (reference :605-717)."""

    def __init__(self, dim, window_size, pretrained_window_size, num_heads, version="swinv1", qkv_bias=False, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, norm=None, **spiking_kwargs):
        super().__init__()
        if dim % num_heads or (dim // num_heads) != 32:
            raise NotImplementedError("the HIP gate kernel is built for head_dim == 32 (all shipped configs)")
        self.dim, self.window_size, self.num_heads, self.norm_layer = dim, tuple(window_size), num_heads, norm
        kw = _neuron_kwargs(dict(spiking_kwargs, num_steps=self.window_size[0]))     # neurons run over T' = Wd (:615)
        n_tok = self.window_size[0] * self.window_size[1] * self.window_size[2]
        self.positional_encoding = nn.Parameter(torch.zeros(1, num_heads, n_tok, dim // num_heads))
        self.linear_q = nn.Linear(dim, dim, bias=False)
        self.bn_q = SpikingNormLayer(dim, self.window_size[0], norm, spiking_kwargs["v_th"])
        self.sn_q = Spiking_neuron(**kw)
        self.linear_k = nn.Linear(dim, dim, bias=False)
        self.bn_k = SpikingNormLayer(dim, self.window_size[0], norm, spiking_kwargs["v_th"])
        self.sn_k = Spiking_neuron(**kw)
        self.sn2_q = Spiking_neuron(**kw)
        self.attn_sn = Spiking_neuron(**kw)          # dead on the forward path (:711), kept for the state_dict
        self.proj = nn.Linear(dim, dim)
        self.proj_bn = SpikingNormLayer(dim, self.window_size[0], norm, spiking_kwargs["v_th"])
        self.proj_sn = Spiking_neuron(**kw)

    def forward(self, x, mask=None):
        from ..module_forward import qk_attention_forward
        return qk_attention_forward(self, x, mask)

    def flops(self, N):
        """MACs of one window of N tokens.  The reference class has no `flops()` (its model-level `flops()` raises for the
        shipped MS model); counted here like its SEW sibling (:377-392) with what this attention really
        does: q and k projections, the projection, their norms - the token gate is additions and a compare, no MACs."""
        return N * self.dim * self.dim * 2 + N * self.dim * 2 + N * self.dim * self.dim + N * self.dim

    def record_flops(self, nW, N):
        d = nW * N * self.dim * self.dim
        return {"q": d, "k": d, "attn": 0, "proj": d}


class Spiking_BN_WindowAttention3D(nn.Module):
    """SEW spiking window attention, swinv1 branch (reference :184-370): spike q,k,v, (q*scale) k^T + table bias
    (+ mask), NO softmax, @ v, projection -> BN -> SN.  forward(x (T',B_,Wh,Ww,C) spikes, mask) -> (B_, N, C) spikes.
    The score / bias / mask / .V core is one fused MFMA kernel (csrc/win_attn.hip)."""

    def __init__(self, dim, window_size, pretrained_window_size, num_heads, version="swinv1", qkv_bias=False, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, norm=None, **spiking_kwargs):
        super().__init__()
        if version != "swinv1":
            raise NotImplementedError("the reference's swinv2 branch references an undefined attribute (:336)")
        if dim // num_heads != 32:
            raise NotImplementedError("the fused window-attention kernel is built for head_dim == 32")
        from ..STSwinNet.swin_transformer3D_v2 import relative_position_index
        self.dim, self.window_size, self.num_heads, self.norm_layer = dim, tuple(window_size), num_heads, norm
        kw = _neuron_kwargs(dict(spiking_kwargs, num_steps=self.window_size[0]))
        self.scale = 1.0 if spiking_kwargs["neuron_type"] in ("psn", "glif") else (qk_scale or (dim // num_heads) ** -0.5)
        wd, wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wd - 1) * (2 * wh - 1) * (2 * ww - 1), num_heads))
        self.register_buffer("relative_position_index", relative_position_index(self.window_size))
        for n in ("q", "k", "v"):
            setattr(self, f"linear_{n}", nn.Linear(dim, dim, bias=False))
            setattr(self, f"bn_{n}", SpikingNormLayer(dim, wd, norm, spiking_kwargs["v_th"]))
            setattr(self, f"sn_{n}", Spiking_neuron(**kw))
        self.attn_sn = Spiking_neuron(**kw)
        self.proj = nn.Linear(dim, dim)
        self.proj_bn = SpikingNormLayer(dim, wd, norm, spiking_kwargs["v_th"])
        self.proj_sn = Spiking_neuron(**kw)

    def forward(self, x, mask=None):
        from .. import hip
        from ..engine import bn_affine, _np
        if self.training:
            raise NotImplementedError("forward-only")
        with torch.no_grad():
            Tq, B_, Wh, Ww, C = x.shape
            N1, dev, nsplit = Wh * Ww, x.device, 2
            M, n = Tq * B_ * N1, B_ * N1 * C
            xs = x.reshape(M, C).float().contiguous()                       # the SEW stream carries SUMS of spikes: dense projections
            spk = {}
            for name in ("q", "k", "v"):
                lin, bn, sn = getattr(self, f"linear_{name}"), getattr(self, f"bn_{name}"), getattr(self, f"sn_{name}")
                a, b = bn_affine(bn.norm_layer, dev)
                f = torch.mm(xs, lin.weight.detach().float().t())
                spk[name] = torch.empty((M, C), dtype=torch.uint8, device=dev)
                hip.neuron_fwd(f, spk[name], Tq, 1, n, 0, n, 0, n, _np(sn, dev), alpha=a, beta=b, Cch=C, inner=1)
            N = Tq * N1
            idx = self.relative_position_index[:N, :N].reshape(-1)
            bias = self.relative_position_bias_table.detach()[idx].reshape(N, N, -1).permute(2, 0, 1).contiguous()
            scale = torch.full((self.num_heads,), float(self.scale), device=dev)
            z = hip.win_attn_sew(spk["q"], spk["k"], spk["v"], scale, bias, None if mask is None else mask.contiguous(),
                                 self.num_heads, Tq, B_, N1)
            y = torch.addmm(self.proj.bias.detach(), z.view(M, C), self.proj.weight.detach().t())   # dense fp32 GEMM (rocBLAS)
            a, b = bn_affine(self.proj_bn.norm_layer, dev)
            out = torch.empty((M, C), dtype=torch.float32, device=dev)
            hip.neuron_fwd(y, out, Tq, 1, n, 0, n, 0, n, _np(self.proj_sn, dev), alpha=a, beta=b, Cch=C, inner=1)
            return out.view(B_, N, C), None

    def flops(self, N):
        """MACs of one window of N tokens (reference :377-392)."""
        f = N * self.dim * self.dim * 3 + N * self.dim * 3
        f += 2 * self.num_heads * N * (self.dim // self.num_heads) * N
        return f + N * self.dim * self.dim + N * self.dim

    def record_flops(self, nW, N):
        """reference :394-411."""
        d = nW * N * self.dim * self.dim
        return {"q": d, "k": d, "v": d, "attn": 2 * nW * self.num_heads * N * (self.dim // self.num_heads) * N, "proj": d}


class MS_Spiking_Mlp(nn.Module):
    """SN -> fc1 -> BN -> SN -> fc2 -> BN (reference :115-181)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, norm_layer="BN", act_layer=None, drop=0.0,
                 **spiking_kwargs):
        super().__init__()
        out_features, hidden_features = out_features or in_features, hidden_features or in_features
        self.norm_layer = norm_layer
        kw = _neuron_kwargs(spiking_kwargs)
        self.fc1 = nn.Linear(in_features, hidden_features, bias=False)
        self.bn1 = SpikingNormLayer(hidden_features, spiking_kwargs["num_steps"], spiking_kwargs["spike_norm"], spiking_kwargs["v_th"])
        self.sn1 = Spiking_neuron(**kw)
        self.fc2 = nn.Linear(hidden_features, out_features, bias=False)
        self.bn2 = SpikingNormLayer(out_features, spiking_kwargs["num_steps"], spiking_kwargs["spike_norm"], spiking_kwargs["v_th"])
        self.sn2 = Spiking_neuron(**kw)

    def forward(self, x):
        from ..module_forward import ms_mlp_forward
        return ms_mlp_forward(self, x)


class MS_Spiking_SwinTransformerBlock3D(nn.Module):
    """x += SSA(x); x += MLP(x) with membrane shortcuts (reference :720-895)."""
    attn_module, mlp_module = None, None                  # set below the class bodies (MS: QK attention + MS MLP)

    def __init__(self, dim, input_resolution, num_heads, window_size=(2, 7, 7), pretrained_window_size=(0, 0, 0),
                 shift_size=(0, 0, 0), mlp_ratio=4.0, version="swinv1", qkv_bias=True, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=None, norm_layer="LN", use_checkpoint=False, **spiking_kwargs):
        super().__init__()
        if norm_layer != "BN":
            raise NotImplementedError("only spike_norm 'BN' (the shipped configs) is supported")
        assert all(0 <= s < w for s, w in zip(shift_size, window_size)), "shift_size must in 0-window_size"
        self.dim, self.input_resolution, self.num_heads = dim, input_resolution, num_heads
        self.window_size, self.shift_size, self.mlp_ratio = tuple(window_size), tuple(shift_size), mlp_ratio
        self.norm_layer, self.drop_path_rate, self.cnf = norm_layer, drop_path, "ADD"
        self.attn = self.attn_module(dim, self.window_size, pretrained_window_size, num_heads, version, qkv_bias,
                                     qk_scale, attn_drop, drop, norm=norm_layer, **spiking_kwargs)
        self.mlp = self.mlp_module(dim, int(dim * mlp_ratio), norm_layer=norm_layer, drop=drop, **spiking_kwargs)

    def forward(self, x, mask_matrix=None, return_attention=False):
        from ..module_forward import ms_block_forward
        return ms_block_forward(self, x, mask_matrix, return_attention)

    def flops(self):
        """Analytic MAC count of the reference (:849-866); the reference's MS block raises here because its QK attention has no
        `flops()` - this build counts that attention's three Linear layers the way the SEW attention counts its own."""
        H, W = self.input_resolution
        nW = H * W // self.window_size[1] // self.window_size[2]
        f = nW * self.attn.flops(self.window_size[0] * self.window_size[1] * self.window_size[2])
        f += 2 * H * W * self.dim * self.dim * self.mlp_ratio
        return f + H * W * self.dim * self.mlp_ratio + H * W * self.dim

    def record_flops(self):
        H, W = self.input_resolution
        nW = H * W // self.window_size[1] // self.window_size[2]
        return {"attn": self.attn.record_flops(nW, self.window_size[0] * self.window_size[1] * self.window_size[2]),
                "mlp0": H * W * self.dim * self.dim * self.mlp_ratio, "mlp1": H * W * self.dim * self.dim * self.mlp_ratio}


class MS_SpikingPatchMerging(nn.Module):
    """2x2 gather -> SN -> Linear 4C->2C -> BN (reference :898-974)."""

    def __init__(self, input_resolution, dim, norm_layer="BN", **spiking_kwargs):
        super().__init__()
        self.input_resolution, self.dim = input_resolution, dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = SpikingNormLayer(2 * dim, spiking_kwargs["num_steps"], norm_layer, spiking_kwargs["v_th"])
        self.sn = Spiking_neuron(**_neuron_kwargs(spiking_kwargs))

    def forward(self, x):
        from ..module_forward import ms_patch_merging_forward
        return ms_patch_merging_forward(self, x)

    def flops(self):
        """reference :936-941."""
        H, W = self.input_resolution
        return (H // 2) * (W // 2) * 4 * self.dim * 2 * self.dim + (H // 2) * (W // 2) * self.dim // 2

    def record_flops(self):
        H, W = self.input_resolution
        return (H // 2) * (W // 2) * 4 * self.dim * 2 * self.dim


class MS_Spiking_Swin_BasicLayer(nn.Module):
    """One stage: `depth` blocks alternating W-MSA / SW-MSA + optional patch merging (reference :995-1129)."""
    swin_block_type = MS_Spiking_SwinTransformerBlock3D

    def __init__(self, dim, input_resolution, depth, num_heads, window_size=(1, 7, 7), pretrained_window_size=(1, 7, 7),
                 mlp_ratio=4.0, version="swinv1", qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0, drop_path=0.0,
                 norm_layer="LN", downsample=None, use_checkpoint=False, **spiking_kwargs):
        super().__init__()
        self.dim, self.input_resolution, self.depth = dim, input_resolution, depth
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        self.swin_blocks = nn.ModuleList([
            self.swin_block_type(
                dim, input_resolution, num_heads, self.window_size, pretrained_window_size,
                (0, 0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio, version, qkv_bias, qk_scale, drop, attn_drop,
                drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer, **spiking_kwargs)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer, **spiking_kwargs) if downsample else None

    def flops(self):
        """reference :1109-1115."""
        return sum(b.flops() for b in self.swin_blocks) + (self.downsample.flops() if self.downsample is not None else 0)

    def record_flops(self):
        rec = {f"block{i}": b.record_flops() for i, b in enumerate(self.swin_blocks)}
        if self.downsample is not None:
            rec["downsample"] = self.downsample.record_flops()
        return rec


class MS_Spiking_SwinTransformer3D_v2(nn.Module):
    """Backbone: patch embedding + stages (reference :1132-1292)."""
    swin_layer_type, downsample_layer_type = MS_Spiking_Swin_BasicLayer, MS_SpikingPatchMerging

    def __init__(self, pretrained=None, pretrained2d=False, arc_type="swinv1", embed_type="PatchEmbedLocal", img_size=(320, 480),
                 patch_size=(4, 4, 4), in_chans=3, embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
                 window_size=(2, 7, 7), pretrained_window_size=(2, 7, 7), mlp_ratio=4.0, qkv_bias=True, qk_scale=0.125,
                 drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.2, norm_layer="BN", patch_norm=False,
                 out_indices=(0, 1, 2, 3), frozen_stages=-1, use_checkpoint=False, norm=None, **spiking_kwargs):
        super().__init__()
        if embed_type != "MS_PED_Spiking_PatchEmbed_Conv_sfn":
            raise NotImplementedError(f"patch embedding {embed_type!r}: only the shipped MS_PED_Spiking_PatchEmbed_Conv_sfn is built")
        self.num_layers, self.embed_dim, self.window_size = len(depths), embed_dim, tuple(window_size)
        self.patch_size, self.out_indices, self.norm_layer = patch_size, tuple(out_indices), norm_layer
        self.patch_embed = MS_PED_Spiking_PatchEmbed_Conv_sfn(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                                              embed_dim=embed_dim, patch_norm=None, norm=norm,
                                                              spiking_proj=True, **spiking_kwargs)
        self.patches_resolution = self.patch_embed.patches_resolution
        dpr = [float(v) for v in np.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            self.layers.append(self.swin_layer_type(
                int(embed_dim * 2 ** i), (self.patches_resolution[0] // 2 ** i, self.patches_resolution[1] // 2 ** i), depths[i],
                num_heads[i], window_size, pretrained_window_size, mlp_ratio, arc_type, qkv_bias, qk_scale, drop_rate,
                attn_drop_rate, dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer,
                self.downsample_layer_type if i < self.num_layers - 1 else None, use_checkpoint, **spiking_kwargs))
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]

    def flops(self):
        """reference :1266-1274: patch embedding + stages."""
        return self.patch_embed.flops() + sum(layer.flops() for layer in self.layers)

    def record_flops(self):
        rec = {"patch_embed": self.patch_embed.record_flops()}
        for i, layer in enumerate(self.layers):
            rec[f"layer{i}"] = layer.record_flops()
        return rec


MS_Spiking_SwinTransformerBlock3D.attn_module = Spiking_QK_WindowAttention3D
MS_Spiking_SwinTransformerBlock3D.mlp_module = MS_Spiking_Mlp


# ---------------------------------------------------------------------------------------------- SEW family (reference :115-162, :720-886, :898-950)
class Spiking_Mlp(MS_Spiking_Mlp):
    """fc1 -> BN -> SN -> fc2 -> BN -> SN (reference :115-162); same parameters as the MS variant, the SEW engine orders them."""

    def forward(self, x):
        from ..module_forward import sew_mlp_forward
        return sew_mlp_forward(self, x)


class Spiking_SwinTransformerBlock3D(MS_Spiking_SwinTransformerBlock3D):
    """x = SSA(x) + x; x = MLP(x) + x with spike-element-wise ADD shortcuts (reference :720-886)."""
    attn_module, mlp_module = Spiking_BN_WindowAttention3D, Spiking_Mlp

    def forward(self, x, mask_matrix=None, return_attention=False):
        from ..module_forward import sew_block_forward
        return sew_block_forward(self, x, mask_matrix, return_attention)


class SpikingPatchMerging(MS_SpikingPatchMerging):
    """2x2 gather -> Linear 4C->2C -> BN -> SN (reference :898-950)."""

    def forward(self, x):
        from ..module_forward import sew_patch_merging_forward
        return sew_patch_merging_forward(self, x)


class Spiking_Swin_BasicLayer(MS_Spiking_Swin_BasicLayer):
    swin_block_type = Spiking_SwinTransformerBlock3D


class Spiking_SwinTransformer3D_v2(MS_Spiking_SwinTransformer3D_v2):
    """SEW backbone (reference :1132-1285)."""
    swin_layer_type, downsample_layer_type = Spiking_Swin_BasicLayer, SpikingPatchMerging
