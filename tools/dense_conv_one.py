#!/usr/bin/env python3
"""A few launches of the dense 3x3 convolution on config 3's patch-embedding shape, for rocprofv3 --pmc passes (tools/pmc_dense_conv.sh)."""
import sys
import torch
sys.path.insert(0, ".")
from sdformerflow_amd import hip
imgs, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 288, 384)
g = torch.Generator().manual_seed(0)
xp = hip.pack_planes(torch.randn(imgs, 96, H, W, generator=g).cuda())
rp = hip.pack_planes(torch.randn(imgs, 96, H, W, generator=g).cuda())
wp = hip.pack_dense_conv_weight((torch.randn(96, 96, 3, 3, generator=g) / 30).cuda())
al, be = (0.5 + torch.rand(96, generator=g)).cuda(), torch.randn(96, generator=g).cuda()
for _ in range(5):
    hip.dense_conv3x3(xp, wp, al, be, rp, True)
torch.cuda.synchronize()
