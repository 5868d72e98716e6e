"""ANN patch embedding on the path of BASELINE config 3 (mirror of reference models/STSwinNet/PatchEmbed.py:104-196,
`PatchEmbedLocal`): per temporal chunk, Conv2d head -> 4 BatchNorm residual blocks -> strided Conv2d projection.
The nine stride-1 convolutions (60 % of config 3 on the library's fp32 Winograd kernel) run on this framework's dense
convolution (csrc/dense_conv_wres.hip: two fp16 planes per operand, BatchNorm / residual / ReLU in the epilogue, activations
kept as hi + lo planes between the layers); the stride-4 projection stays a library convolution on the channels-last result."""
import os

import torch
import torch.nn as nn

from .. import hip


class ResidualBlock(nn.Module):
    """conv3x3 - [BN] - relu - conv3x3 - [BN], + input, relu (reference models/submodules.py:160-229).
    With a norm the convolutions carry no bias, exactly as the reference builds them."""

    def __init__(self, in_channels, out_channels, norm=None):
        super().__init__()
        bias = norm != "BN"
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=bias)
        if norm == "BN":
            self.bn1 = nn.BatchNorm2d(out_channels)
            self.bn2 = nn.BatchNorm2d(out_channels)
        self.norm = norm
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1, bias=bias)

    def forward(self, x):
        y = self.conv1(x)
        if self.norm == "BN":
            y = self.bn1(y)
        y = self.conv2(torch.relu(y))
        if self.norm == "BN":
            y = self.bn2(y)
        return torch.relu(y + x)


class _ResidualEncoding(nn.Module):
    def __init__(self, dim):
        super().__init__()
        for i in range(1, 5):
            self.add_module(f"resblock{i}", ResidualBlock(dim, dim, norm="BN"))

    def forward(self, x):
        for i in range(1, 5):
            x = getattr(self, f"resblock{i}")(x)
        return x


class PatchEmbedLocal(nn.Module):
    """forward(x (T,B,bins/T,H,W)) -> (B, embed_dim, T, H/ph, W/pw)."""

    def __init__(self, img_size=(240, 320), patch_size=(10, 4, 4), in_chans=20, embed_dim=96):
        super().__init__()
        self.patch_size = tuple(patch_size)
        self.num_blocks = in_chans // patch_size[0]
        self.patches_resolution = [img_size[0] // patch_size[1], img_size[1] // patch_size[2]]
        self.embed_dim = embed_dim
        self.head = nn.Conv2d(patch_size[0], embed_dim, 3, 1, 1)
        self.residual_encoding = _ResidualEncoding(embed_dim)
        self.proj = nn.Conv2d(embed_dim, embed_dim, 3, self.patch_size[1:], 1)

    def _packed(self):
        """fp16 weight planes and folded eval-BatchNorm (alpha, beta) of the stride-1 convolutions, rebuilt when a tensor changes."""
        tensors = list(self.head.parameters()) + list(self.residual_encoding.parameters()) + list(self.residual_encoding.buffers())
        stamp = tuple((t.data_ptr(), t._version) for t in tensors)
        if getattr(self, "_pk_stamp", None) != stamp:
            def fold(bn):
                a = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
                return a.contiguous(), (bn.bias.detach().float() - bn.running_mean.float() * a).contiguous()
            pk = [(hip.pack_dense_conv_weight(self.head.weight), None, self.head.bias.detach().float().contiguous())]
            for i in range(1, 5):
                rb = getattr(self.residual_encoding, f"resblock{i}")
                pk.append((hip.pack_dense_conv_weight(rb.conv1.weight),) + fold(rb.bn1))
                pk.append((hip.pack_dense_conv_weight(rb.conv2.weight),) + fold(rb.bn2))
            self._pk, self._pk_stamp = pk, stamp
        return self._pk

    def _packed_proj(self):
        w = self.proj.weight
        stamp = (w.data_ptr(), w._version)
        if getattr(self, "_pkp_stamp", None) != stamp:
            self._pkp = hip.pack_dense_linear_weight(w.detach().float().permute(0, 2, 3, 1).reshape(w.shape[0], -1))
            self._pkp_stamp = stamp
        return self._pkp

    def _encode_planes(self, x):
        """head + residual encoding of (imgs, C, H, W) fp32 -> (imgs, H, W, embed_dim) fp32, channels last."""
        pk = self._packed()
        a = hip.dense_conv3x3(hip.pack_planes(x), *pk[0])
        for i in range(4):
            y = hip.dense_conv3x3(a, *pk[1 + 2 * i], None, True)
            a = hip.dense_conv3x3(y, *pk[2 + 2 * i], a, True, i == 3)
        return a

    def forward(self, x):
        T, B = x.shape[:2]
        x = x.flatten(0, 1)                                                           # the T chunks share every weight
        imgs, C, H, W = x.shape
        if (not self.training and x.is_cuda and hip.sw("SDF_DENSE_CONV", "1") != "0"
                and hip.dense_conv_applicable(imgs, H, W, C, self.embed_dim) and C <= 16 and self.embed_dim == 96):
            a = self._encode_planes(x)                                                # (imgs, H, W, C) channels last
            st = self.proj.stride
            if st[0] == st[1] and self.embed_dim % 32 == 0 and self.proj.bias is not None:
                # the strided projection as a GEMM over gathered rows; images leave (b, t)-major, channels last: the returned
                # (B, C, T, h, w) is a view of exactly the (B, T, h, w, C) buffer the swin stages ask for (their permute + contiguous
                # is then free)
                y = hip.dense_conv3x3_strided(a, self._packed_proj(), self.proj.bias.detach().float(), st[0], T)
                return y.view(B, T, *y.shape[1:]).permute(0, 4, 1, 2, 3)
            y = self.proj(a.permute(0, 3, 1, 2)).contiguous()
        else:
            y = self.proj(self.residual_encoding(self.head(x)))
        return y.view(T, B, *y.shape[1:]).permute(1, 2, 0, 3, 4)
