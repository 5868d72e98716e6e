"""ANN patch embedding on the path of BASELINE config 3 (mirror of reference models/STSwinNet/PatchEmbed.py:104-196,
`PatchEmbedLocal`): per temporal chunk, Conv2d head -> 4 BatchNorm residual blocks -> strided Conv2d projection.
Dense fp32 convolutions: MIOpen through torch (library convolutions, not a hot op of this framework)."""
import torch
import torch.nn as nn


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

    def forward(self, x):
        T, B = x.shape[:2]
        y = self.proj(self.residual_encoding(self.head(x.flatten(0, 1))))            # the T chunks share every weight
        return y.view(T, B, *y.shape[1:]).permute(1, 2, 0, 3, 4)
