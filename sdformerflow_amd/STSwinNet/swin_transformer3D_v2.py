"""ANN video-swin-v2 pieces on the hot path (mirror of reference models/STSwinNet/swin_transformer3D_v2.py):
`window_partition` :37-49, `window_reverse` :52-65, `get_window_size` :68-81, `compute_mask` :409-421 and the cosine
`WindowAttention3D` :87-205, whose score / bias / mask / softmax / .V core runs in one fused MFMA kernel
(csrc/win_attn.hip) and whose projections - like the Mlp's - run on this framework's Linear kernel (csrc/dense_linear.hip:
two fp16 planes per operand, bias / GELU / shortcut add in the epilogue; `SDF_DENSE_LINEAR=0` keeps the library GEMMs)."""
import math
import os
from functools import lru_cache

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from ..STSwinNet_SNN.Spiking_swin_transformer3D import get_window_size  # noqa: F401  (same function, one definition)


def linear_rows(lin, a, gelu=False, resid=None):
    """act(lin(a)) (+ resid) on rows a (M, K) fp32: one launch of sdf_dense_linear_fwd when the layer has a kernel form (eval, GPU,
    N % 96 == 0, K % 32 == 0), the library's F.linear / F.gelu / add otherwise.  The fp16 weight planes are made once per
    parameter version."""
    M, K = a.shape
    N = lin.out_features
    if (not lin.training and a.is_cuda and a.dtype == torch.float32 and hip.sw("SDF_DENSE_LINEAR", "1") != "0"
            and hip.dense_linear_applicable(M, N, K)):
        stamp = (lin.weight.data_ptr(), lin.weight._version)
        if getattr(lin, "_pk_stamp", None) != stamp:
            lin._pk, lin._pk_stamp = hip.pack_dense_linear_weight(lin.weight), stamp
        bias = lin.bias.detach() if lin.bias is not None else None
        return hip.dense_linear(a.contiguous(), lin._pk, bias, gelu, resid)
    y = F.linear(a, lin.weight, lin.bias)
    if gelu:
        y = F.gelu(y)
    return y if resid is None else resid + y


def layer_norm(ln, x):
    """nn.LayerNorm `ln` on x (..., C): the HIP kernel in eval on the GPU (one read, one write), torch otherwise."""
    C = x.shape[-1]
    if (not ln.training and x.is_cuda and x.dtype == torch.float32 and ln.elementwise_affine and ln.bias is not None and C % 4 == 0
            and C <= 2048 and hip.sw("SDF_LAYER_NORM", "1") != "0"):
        return hip.layer_norm(x.contiguous(), ln.weight.detach(), ln.bias.detach(), ln.eps)
    return ln(x)


def window_partition(x, window_size):
    """(B,D,H,W,C) -> (B*nW, Wd*Wh*Ww, C), windows ordered (b, d-block, h-block, w-block)."""
    B, D, H, W, C = x.shape
    wd, wh, ww = window_size
    x = x.reshape(B, D // wd, wd, H // wh, wh, W // ww, ww, C)
    return x.permute(0, 1, 3, 5, 2, 4, 6, 7).reshape(-1, wd * wh * ww, C)


def window_reverse(windows, window_size, B, D, H, W):
    """Inverse of window_partition: (B*nW, Wd, Wh, Ww, C) -> (B,D,H,W,C)."""
    wd, wh, ww = window_size
    x = windows.reshape(B, D // wd, H // wh, W // ww, wd, wh, ww, -1)
    return x.permute(0, 1, 4, 2, 5, 3, 6, 7).reshape(B, D, H, W, -1)


@lru_cache()
def compute_mask(D, H, W, window_size, shift_size, device):
    """0 / -100 mask (nW, N, N) of the 27 shifted regions."""
    def labels(n, w, s):
        r = np.zeros(n, np.int64)
        idx = np.arange(n)
        for c, sl in enumerate((slice(-w), slice(-w, -s), slice(-s, None))):
            r[idx[sl]] = c
        return r
    rd, rh, rw = (labels(n, w, s) for n, w, s in zip((D, H, W), window_size, shift_size))
    img = torch.from_numpy((rd[:, None, None] * 9 + rh[None, :, None] * 3 + rw[None, None, :]).astype(np.float32))
    mw = window_partition(img.view(1, D, H, W, 1), window_size).squeeze(-1)
    diff = mw.unsqueeze(1) - mw.unsqueeze(2)
    return torch.where(diff != 0, torch.tensor(-100.0), torch.tensor(0.0)).to(device)


def relative_position_index(ws):
    wd, wh, ww = ws
    c = np.stack(np.meshgrid(np.arange(wd), np.arange(wh), np.arange(ww), indexing="ij")).reshape(3, -1)
    rel = c[:, :, None] - c[:, None, :]
    return torch.from_numpy((rel[0] + wd - 1) * (2 * wh - 1) * (2 * ww - 1) + (rel[1] + wh - 1) * (2 * ww - 1)
                            + rel[2] + ww - 1)


def relative_coords_table(ws, pws):
    """(1, 2Wd-1, 2Wh-1, 2Ww-1, 3) log-spaced coordinates, INCLUDING the reference's axis quirk: the
    normalisation `table[:, :, :, i] /= (ws[i]-1)` (:121-127) addresses the w-offset axis, not the coordinate."""
    ax = [np.arange(-(w - 1), w, dtype=np.float32) for w in ws]
    t = np.stack(np.meshgrid(*ax, indexing="ij"), -1)[None].copy()
    for i in range(3):
        t[:, :, :, i] /= np.float32((pws[i] - 1) if pws[0] > 0 else (ws[i] - 1))
    t *= 8
    return torch.from_numpy((np.sign(t) * np.log2(np.abs(t) + 1.0) / np.log2(8)).astype(np.float32))


class WindowAttention3D(nn.Module):
    """Cosine window attention with continuous relative-position bias (reference :87-205).
    forward(x (B_,N,C), mask (nW,N,N)|None) -> (out (B_,N,C), None); head_dim must be 32."""

    def __init__(self, dim, window_size, pretrained_window_size, num_heads, qkv_bias=False, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0):
        super().__init__()
        if dim // num_heads != 32:
            raise NotImplementedError("the fused window-attention kernel is built for head_dim == 32")
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.pretrained_window_size = tuple(pretrained_window_size)
        self.logit_scale = nn.Parameter(torch.log(10 * torch.ones((num_heads, 1, 1))))
        self.cpb_mlp = nn.Sequential(nn.Linear(3, 512, bias=True), nn.ReLU(inplace=True), nn.Linear(512, num_heads, bias=False))
        self.register_buffer("relative_coords_table", relative_coords_table(self.window_size, self.pretrained_window_size))
        self.register_buffer("relative_position_index", relative_position_index(self.window_size))
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def position_bias(self):
        """16 * sigmoid(cpb_mlp(table)[index]) -> (nH, N, N) (:184-189)."""
        N = self.relative_position_index.shape[0]
        tab = self.cpb_mlp(self.relative_coords_table).view(-1, self.num_heads)
        b = tab[self.relative_position_index.view(-1)].view(N, N, -1).permute(2, 0, 1)
        return (16 * torch.sigmoid(b)).contiguous()

    def _bias_and_scale(self):
        """(position bias (nH, N, N), clamped exp logit scale (nH,)): functions of the parameters alone, so in eval they are computed
        once per parameter version instead of once per forward (ten cpb_mlp evaluations per config-3 forward otherwise)."""
        ps = list(self.cpb_mlp.parameters()) + [self.logit_scale]
        stamp = tuple((p.data_ptr(), p._version) for p in ps)
        if self.training or getattr(self, "_bs_stamp", None) != stamp:
            scale = torch.clamp(self.logit_scale, max=math.log(1.0 / 0.01)).exp().reshape(-1).contiguous()
            self._bs, self._bs_stamp = (self.position_bias(), scale), stamp
        return self._bs

    def forward(self, x, mask=None):
        if self.training:
            raise NotImplementedError("forward-only (SURVEY.md section 8f row 3 covers the backward kernels)")
        with torch.no_grad():
            qkv = self.qkv(x).contiguous()
            bias, scale = self._bias_and_scale()
            o = hip.win_attn_ann(qkv, scale, bias, None if mask is None else mask.contiguous(), self.num_heads)
            return self.proj(o), None

    def forward_rows(self, y2, row_map, B_, mask, resid=None, norm=None):
        """Attention on un-partitioned rows y2 (rows, C): windows are formed by `row_map` inside the kernel -> (rows, C),
        plus `resid` (the block's shortcut, added in the projection's epilogue).  y2 None: `resid` is the block input x and `norm`
        its LayerNorm - the whole half block x + proj(attention(norm(x))) as one launch (csrc/ann_block.hip)."""
        with torch.no_grad():
            if y2 is None:
                stamp = (self.qkv.weight.data_ptr(), self.qkv.weight._version, self.proj.weight.data_ptr(), self.proj.weight._version)
                if getattr(self, "_blk_stamp", None) != stamp:
                    self._blk, self._blk_stamp = hip.pack_ann_attn_block_weights(self.qkv.weight, self.proj.weight, self.num_heads), stamp
                bias, scale = self._bias_and_scale()
                # (keyed by the parameter versions the bias was made from - not by its address, which a later bias may be given again)
                tkey = (self._bs_stamp, None if mask is None else (mask.data_ptr(), mask._version))
                if getattr(self, "_tab_key", None) != tkey:               # (bias + mask) * log2 e: once per parameter version and mask
                    self._tab, self._tab_key = hip.ann_attn_block_table(scale, bias, mask), tkey
                    self._tab_keep = (bias, mask)                          # (the key's pointers stay valid while the table does)
                N = self.window_size[0] * self.window_size[1] * self.window_size[2]
                x2 = resid.contiguous()
                return hip.ann_attn_block(x2, torch.empty_like(x2), row_map, B_, N, self.num_heads, norm.weight.detach(), norm.bias.detach(),
                                          norm.eps, self._blk[0], None if self.qkv.bias is None else self.qkv.bias.detach(), self._tab[0],
                                          self._tab[1], self._blk[1], None if self.proj.bias is None else self.proj.bias.detach())
            qkv = linear_rows(self.qkv, y2)
            pad = self.qkv.bias.detach().float().contiguous() if self.qkv.bias is not None else torch.zeros(3 * self.dim, device=y2.device)
            bias, scale = self._bias_and_scale()
            N = self.window_size[0] * self.window_size[1] * self.window_size[2]
            o = hip.win_attn_ann_windowed(qkv, row_map, B_, N, pad, scale, bias, mask, self.num_heads)
            return linear_rows(self.proj, o, False, resid)


class Mlp(nn.Module):
    """fc1 -> GELU -> fc2 (reference :15-34; dropout is identity in eval)."""

    def __init__(self, in_features, hidden_features=None, out_features=None):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)

    def forward(self, x, resid=None):
        """fc2(gelu(fc1(x))) (+ resid): GELU rides in fc1's epilogue, the shortcut in fc2's."""
        shp = x.shape
        h = linear_rows(self.fc1, x.reshape(-1, shp[-1]), True)
        y = linear_rows(self.fc2, h, False, None if resid is None else resid.reshape(-1, resid.shape[-1]))
        return y.view(*shp[:-1], y.shape[-1])


class SwinTransformerBlock3D(nn.Module):
    """Pre-norm video-swin block (reference :208-336): x + attn(LN(x)) over (shifted) 3-D windows, then x + MLP(LN(x)).
    Input and output (B,D,H,W,C)."""

    def __init__(self, dim, num_heads, window_size=(2, 7, 7), shift_size=(0, 0, 0), mlp_ratio=4.0, qkv_bias=True,
                 pretrained_window_size=(0, 0, 0)):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.window_size, self.shift_size = tuple(window_size), tuple(shift_size)
        self.norm1 = nn.LayerNorm(dim)
        self.attn = WindowAttention3D(dim, self.window_size, pretrained_window_size, num_heads, qkv_bias=qkv_bias)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    _maps = {}             # (B, D, H, W, window, shift, device) -> device row map, shared by the blocks of a process; LRU-bounded
    _MAPS_MAX = 32

    def forward(self, x, mask_matrix=None):
        """forward_part1 + forward_part2 (reference :272-313).  The reference pads, rolls and partitions LN(x) into windows and
        reverses all of that behind the attention (:286-310); here the qkv projection runs on the un-partitioned rows and the
        attention kernel reads / writes them through the slice map (index arithmetic inside the kernel, nothing materialised):
        the projection is row-wise, so it commutes with the permutation; a padding token is a zero row, whose q | k | v is the
        qkv bias.  `SDF_ATTN_MATERIALISE=1` keeps the reference's sequence as the A/B path."""
        B, D, H, W, C = x.shape
        ws, ss = get_window_size((D, H, W), self.window_size, self.shift_size)
        if tuple(ws) != tuple(self.window_size):
            # the reference clamps the window to the feature map (:277) and then fails to add its (N, N) position bias, built for
            # the nominal window, to the smaller scores (:190, RuntimeError); the same input is refused here, not mis-addressed
            raise RuntimeError(f"feature map {(D, H, W)} is smaller than the window {self.window_size}: the relative position bias of "
                               "WindowAttention3D is defined for the nominal window only (reference swin_transformer3D_v2.py:184-190)")
        materialise = hip.sw("SDF_ATTN_MATERIALISE") == "1" or self.training
        # first stage (C = 96, three heads, 162-token windows): LayerNorm -> qkv -> attention -> proj -> + x is ONE launch that reads x
        # through the slice map (csrc/ann_block.hip); elsewhere the norm runs here and forward_rows takes its output
        fused = (not materialise and x.is_cuda and x.dtype == torch.float32 and self.norm1.elementwise_affine and self.norm1.bias is not None
                 and hip.ann_attn_block_supported(C, self.num_heads, ws[0] * ws[1] * ws[2]) and x.numel() * 4 < 1 << 31)
        y = None if fused else layer_norm(self.norm1, x)
        Dp, Hp, Wp = D + (-D) % ws[0], H + (-H) % ws[1], W + (-W) % ws[2]
        shifted = any(s > 0 for s in ss)
        mask = (mask_matrix if mask_matrix is not None else compute_mask(Dp, Hp, Wp, ws, ss, x.device)) if shifted else None
        if materialise:
            y = F.pad(y, (0, 0, 0, Wp - W, 0, Hp - H, 0, Dp - D))
            if shifted:
                y = torch.roll(y, shifts=(-ss[0], -ss[1], -ss[2]), dims=(1, 2, 3))
            a, _ = self.attn(window_partition(y, ws), mask)
            a = window_reverse(a, ws, B, Dp, Hp, Wp)
            if shifted:
                a = torch.roll(a, shifts=ss, dims=(1, 2, 3))
            x = x + a[:, :D, :H, :W]
        else:
            key = (B, D, H, W, ws, ss, str(x.device))
            maps = SwinTransformerBlock3D._maps
            if key not in maps:
                while len(maps) >= SwinTransformerBlock3D._MAPS_MAX:      # bounded: evaluation sweeps over many input sizes
                    maps.pop(next(iter(maps)))                          # must not grow GPU memory without limit (oldest first)
                maps[key] = hip.window_slice_map(B, D, H, W, ws, ss, x.device)
            else:
                maps[key] = maps.pop(key)                               # most recently used last
            row_map, B_ = maps[key]
            x = self.attn.forward_rows(y.reshape(-1, C) if y is not None else None, row_map, B_, None if mask is None else mask.contiguous(),
                                       x.reshape(-1, C), norm=self.norm1).view(B, D, H, W, C)
        # first stage (C = 96, hidden 384): LayerNorm -> fc1 -> GELU -> fc2 -> + x is one launch, the hidden activations stay in registers
        # (csrc/ann_mlp_block.hip); elsewhere LayerNorm, fc1 (+ GELU) and fc2 (+ shortcut) are three
        if (not self.training and x.is_cuda and x.dtype == torch.float32 and self.norm2.elementwise_affine and self.norm2.bias is not None
                and hip.ann_mlp_block_supported(C, self.mlp.fc1.out_features) and x.numel() * 4 < 1 << 31):
            with torch.no_grad():
                m = self.mlp
                stamp = (m.fc1.weight.data_ptr(), m.fc1.weight._version, m.fc2.weight.data_ptr(), m.fc2.weight._version)
                if getattr(self, "_mlp_stamp", None) != stamp:
                    self._mlp_pk, self._mlp_stamp = hip.pack_ann_mlp_block_weights(m.fc1.weight, m.fc2.weight), stamp
                x2 = x.reshape(-1, C).contiguous()
                y = hip.ann_mlp_block(x2, torch.empty_like(x2), self.norm2.weight.detach(), self.norm2.bias.detach(), self.norm2.eps, self._mlp_pk[0],
                                      None if m.fc1.bias is None else m.fc1.bias.detach(), self._mlp_pk[1],
                                      None if m.fc2.bias is None else m.fc2.bias.detach())
            return y.view(B, D, H, W, C)
        return self.mlp(layer_norm(self.norm2, x), x)


class PatchMerging(nn.Module):
    """2x2 spatial concat -> LN(4C) -> Linear(4C, 2C, no bias) (reference :343-393)."""

    def __init__(self, dim):
        super().__init__()
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = nn.LayerNorm(4 * dim)

    def forward(self, x):
        H, W = x.shape[2:4]
        if H % 2 or W % 2:
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
        y = layer_norm(self.norm, x)
        return linear_rows(self.reduction, y.reshape(-1, y.shape[-1])).view(*y.shape[:-1], -1)


class Swin_BasicLayer(nn.Module):
    """One stage (reference :424-512): `depth` blocks alternating plain / shifted windows, then the optional merge.
    forward((B,D,H,W,C)) -> (stage output, merged output)."""

    def __init__(self, dim, depth, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True, downsample=None,
                 pretrained_window_size=(0, 0, 0)):
        super().__init__()
        self.window_size = tuple(window_size)
        self.shift_size = tuple(i // 2 for i in window_size)
        self.swin_blocks = nn.ModuleList([
            SwinTransformerBlock3D(dim, num_heads, self.window_size, (0, 0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio,
                                   qkv_bias, pretrained_window_size) for i in range(depth)])
        self.downsample = downsample(dim) if downsample is not None else None

    def forward(self, x):
        for blk in self.swin_blocks:
            x = blk(x)
        return x, (self.downsample(x) if self.downsample is not None else x)


class SwinTransformer3D_v2(nn.Module):
    """Backbone (reference :538-763): patch embedding -> stages; every stage output goes through its own LayerNorm
    `norm{i}` and is returned channel-first (B,C,D,h,w)."""

    def __init__(self, patch_embed, embed_dim, depths, num_heads, window_size, mlp_ratio=4.0, qkv_bias=True,
                 out_indices=(0, 1, 2), pretrained_window_size=(0, 0, 0)):
        super().__init__()
        self.patch_embed = patch_embed
        self.num_layers, self.out_indices = len(depths), tuple(out_indices)
        self.num_features = [embed_dim * 2 ** i for i in range(self.num_layers)]
        self.layers = nn.ModuleList([
            Swin_BasicLayer(self.num_features[i], depths[i], num_heads[i], window_size, mlp_ratio, qkv_bias,
                            PatchMerging if i < self.num_layers - 1 else None, pretrained_window_size)
            for i in range(self.num_layers)])
        for i in self.out_indices:
            self.add_module(f"norm{i}", nn.LayerNorm(self.num_features[i]))

    def forward(self, x):
        x = self.patch_embed(x).permute(0, 2, 3, 4, 1).contiguous()                 # (B,D,h,w,C)
        outs = []
        for i, layer in enumerate(self.layers):
            o, x = layer(x)
            if i in self.out_indices:
                outs.append(layer_norm(getattr(self, f"norm{i}"), o).permute(0, 4, 1, 2, 3))
        return outs
