"""ANN counterpart of the flow network (BASELINE config 3; mirror of reference models/STSwinNet/STSwinNet.py):
`STT_encoder` :14-138, `STT_MultiResUNet` :140-283, `STTFlowNet` :309-481, `STTFlowNet_4en` :486-497.
The window-attention core (section 8 row a10) is this framework's fused HIP kernel; the dense glue around it
(LayerNorm, Linear, GELU, fp32 convolutions, bilinear upsampling) is library work through torch on the GPU.
Forward-only: the modules refuse training mode and CPU tensors (no fallback path)."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import hip
from .PatchEmbed import PatchEmbedLocal, ResidualBlock
from .swin_transformer3D_v2 import SwinTransformer3D_v2


class ConvLayer(nn.Module):
    """conv2d + optional relu (reference models/submodules.py:14-67, norm None)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, activation="relu"):
        super().__init__()
        self.conv2d = nn.Conv2d(in_channels, out_channels, kernel_size, stride, kernel_size // 2)
        self.activation = activation

    def forward(self, x):
        y = self.conv2d(x)
        return torch.relu(y) if self.activation == "relu" else y


class UpsampleConvLayer(ConvLayer):
    """bilinear x2 (align_corners False) -> conv2d -> relu (reference models/submodules.py:117-157).
    `forward_parts` is the same layer on the channel concatenation of `parts` without materialising it: each part is upsampled
    straight into its records of one activation-planes tensor (sdf_pack_planes_up2) and the 3x3 convolution runs on this
    framework's dense convolution as a chain of 96-channel slices (hip.dense_conv3x3_wide); `order` says where each part's
    channels sit in the layer's weight (the reference concatenates [prediction, features, skip], the planes hold the
    2-channel prediction last so that every slice starts on a 16-channel record)."""

    def parts_plan(self, parts, order):
        """Record layout of the parts, or None when this layer / these shapes have no dense-convolution form."""
        conv = self.conv2d
        if (self.training or conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.out_channels % 32 or self.activation != "relu"
                or hip.sw("SDF_DENSE_CONV", "1") == "0" or not all(p.is_cuda and p.dtype == torch.float32 for p in parts)):
            return None
        if any(p.shape[1] % 16 for p in parts[:-1]):
            return None
        recs = [-(-p.shape[1] // 16) for p in parts]
        slices = hip.dense_conv_slices(sum(recs))
        imgs, _, h, w = parts[0].shape
        if slices is None or not hip.dense_conv_applicable(imgs, 2 * h, 2 * w, 96, conv.out_channels) \
                or imgs * 4 * h * w * sum(recs) * 64 >= 1 << 31:
            return None
        return recs, slices

    def _packed(self, parts, order, recs, slices):
        conv = self.conv2d
        stamp = (conv.weight.data_ptr(), conv.weight._version, conv.bias.data_ptr(), conv.bias._version, tuple(order), tuple(recs))
        if getattr(self, "_pk_stamp", None) != stamp:
            w = conv.weight.detach().float()
            start = [0] * len(parts)                                   # first weight channel of each part: `order` lists the parts as the
            c = 0                                                      # reference concatenates them
            for i in order:
                start[i] = c
                c += parts[i].shape[1]
            cols = []
            for i, p in enumerate(parts):
                blk = w[:, start[i]:start[i] + p.shape[1]]
                pad = recs[i] * 16 - p.shape[1]
                cols.append(F.pad(blk, (0, 0, 0, 0, 0, pad)) if pad else blk)
            wr = torch.cat(cols, dim=1)                                # weight in the planes' channel order, parts padded to whole records
            self._pk = [((r0, n), hip.pack_dense_conv_weight(wr[:, 16 * r0:16 * (r0 + n)])) for r0, n in slices]
            self._pk_bias = conv.bias.detach().float().contiguous()
            self._pk_stamp = stamp
        return self._pk, self._pk_bias

    def forward_parts(self, parts, order):
        plan = self.parts_plan(parts, order)
        if plan is None:
            return self.forward(torch.cat([parts[i] for i in order], dim=1))
        recs, slices = plan
        imgs, _, h, w = parts[0].shape
        planes = torch.empty((imgs, sum(recs), 2 * h, 2 * w, 32), dtype=torch.float16, device=parts[0].device)
        r0 = 0
        for p, r in zip(parts, recs):
            hip.pack_planes_up2(p, planes, r0)
            r0 += r
        wsl, bias = self._packed(parts, order, recs, slices)
        return hip.dense_conv3x3_wide(planes, wsl, bias, True, True).permute(0, 3, 1, 2)     # (imgs, Cout, 2h, 2w), channels last

    def forward(self, x):
        return super().forward(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False))


def _fit_to(x1, ref):
    """x1 (B, C, h, w) centre-padded (cropped where negative) to ref's spatial size: the ZeroPad2d of `skip_concat`."""
    dY, dX = ref.shape[-2] - x1.shape[-2], ref.shape[-1] - x1.shape[-1]
    return F.pad(x1, (dX // 2, dX - dX // 2, dY // 2, dY - dY // 2)) if dY or dX else x1


class STT_encoder(nn.Module):
    """Swin backbone + one 1x1 projection per (stage, time block) whose outputs are concatenated along channels."""

    def __init__(self, img_size, patch_size, in_chans, embed_dim, depths, num_heads, window_size, pretrained_window_size,
                 mlp_ratio, out_indices):
        super().__init__()
        pe = PatchEmbedLocal(img_size, patch_size, in_chans, embed_dim)
        self.num_blocks = pe.num_blocks
        self.swin3d = SwinTransformer3D_v2(pe, embed_dim, depths, num_heads, window_size, mlp_ratio, True, out_indices,
                                           pretrained_window_size)
        self.projections = nn.ModuleList([
            nn.ModuleList([nn.Conv2d(embed_dim * 2 ** i, embed_dim * 2 ** i // self.num_blocks, 1) for _ in range(self.num_blocks)])
            for i in range(len(depths))])

    def forward(self, x):
        feats = self.swin3d(x)
        outs = []
        for f, ps in zip(feats, self.projections):
            B, _, _, h, w = f.shape
            outs.append(torch.cat([p(c.reshape(B, -1, h, w)) for c, p in zip(f.chunk(self.num_blocks, 2), ps)], dim=1))
        return outs


class STT_MultiResUNet(nn.Module):
    """Encoder -> 2 residual blocks -> decoders with concat skips and a flow prediction per scale."""

    def __init__(self, unet_kwargs, stt_kwargs):
        super().__init__()
        base, E = unet_kwargs["base_num_channels"], unet_kwargs["num_encoders"]
        k, n_out = unet_kwargs["kernel_size"], unet_kwargs["num_output_channels"]
        if unet_kwargs.get("norm") is not None or not unet_kwargs.get("use_upsample_conv", True):
            raise NotImplementedError("only the shipped configuration (norm Null, upsample-conv decoders) is built")
        depths = list(stt_kwargs["swin_depths"])
        if len(depths) != E:
            raise ValueError("swin_depths must list one depth per encoder")
        self.num_encoders = E
        self.encoders = STT_encoder(stt_kwargs.get("input_size", [240, 320]), stt_kwargs["swin_patch_size"], unet_kwargs["num_bins"],
                                    stt_kwargs.get("base_num_channels", base), depths, stt_kwargs["swin_num_heads"],
                                    stt_kwargs["window_size"], stt_kwargs.get("pretrained_window_size", [0, 0, 0]),
                                    stt_kwargs.get("mlp_ratio", 4), stt_kwargs["swin_out_indices"])
        sizes = [base * 2 ** i for i in range(E)]
        top = sizes[-1]
        self.resblocks = nn.ModuleList([ResidualBlock(top, top) for _ in range(unet_kwargs["num_residual_blocks"])])
        self.decoders, self.preds = nn.ModuleList(), nn.ModuleList()
        ins = list(reversed(sizes))
        outs = list(reversed([base] + sizes[:-1]))
        for i, (ci, co) in enumerate(zip(ins, outs)):
            self.decoders.append(UpsampleConvLayer(2 * ci + (0 if i == 0 else n_out), co, k))
            self.preds.append(ConvLayer(co, n_out, 1, activation=None))

    def forward(self, x):
        blocks = self.encoders(x)
        y = blocks[-1]
        for rb in self.resblocks:
            y = rb(y)
        preds = []
        for i, (dec, pred) in enumerate(zip(self.decoders, self.preds)):
            # the reference concatenates [previous prediction, features, skip] (models/STSwinNet/STSwinNet.py:259-283)
            # with `skip_concat` (models/model_util.py:14-19): what is concatenated in front of the skip is centre-padded - cropped
            # where the difference is negative - to the skip's size (odd feature sizes)
            skip = blocks[self.num_encoders - 1 - i]
            parts = [_fit_to(y, skip), skip] + ([_fit_to(preds[-1], skip)] if i > 0 else [])
            y = dec.forward_parts(parts, [2, 0, 1] if i > 0 else [0, 1])
            preds.append(pred(y))
        return preds, None


class STTFlowNet(nn.Module):
    """forward(event_voxel (B,num_bins,H,W), event_cnt, log=False) -> {"flow": [..(B,2,H,W)], "attn": None,
    "spiking_rates": None}; same constructor dictionaries as the reference (`model:` and `swin_transformer:` yaml blocks)."""
    num_en = 3

    def __init__(self, unet_kwargs, stt_kwargs):
        super().__init__()
        unet_kwargs = dict(unet_kwargs)
        self.encoding = unet_kwargs["encoding"]
        self.num_bins = unet_kwargs["num_bins"]
        self.norm_input = unet_kwargs.get("norm_input", False)
        self.mask = unet_kwargs["mask_output"]
        self.num_encoders = self.num_en
        self.num_split = self.num_bins // stt_kwargs["swin_patch_size"][0]
        unet_kwargs.update(num_encoders=self.num_en, num_residual_blocks=2, num_output_channels=2)
        self.sttmultires_unet = STT_MultiResUNet(unet_kwargs, stt_kwargs)

    def detach_states(self):
        pass

    def reset_states(self):
        pass

    def invalidate_engine(self):
        """Drop every packed-weight cache of the tree (fp16 planes of the Linear / convolution layers, position bias).  The caches
        re-pack by themselves on (data_ptr, version) changes; writes through `p.data.copy_()` (EMA swaps) bump neither - call this."""
        from .swin_transformer3D_v2 import SwinTransformerBlock3D
        for m in self.modules():
            for name in ("_pk_stamp", "_pkp_stamp", "_bs_stamp"):
                if hasattr(m, name):
                    setattr(m, name, None)
        SwinTransformerBlock3D._maps.clear()

    @staticmethod
    def normalize(x):
        nz = x != 0
        mean, std = x[nz].mean(), x[nz].std()
        if std > 0:
            x[nz] = (x[nz] - mean) / std
        return x

    def forward(self, event_voxel, event_cnt=None, log=False):
        if self.training:
            raise NotImplementedError("forward-only (SURVEY.md section 8f row 3 covers the backward kernels)")
        if self.encoding == "voxel":
            x = event_voxel
        elif self.encoding == "cnt":
            x = event_cnt
        else:
            raise AttributeError("Model error: Incorrect input encoding.")
        if not x.is_cuda:
            raise hip.SdfError("STTFlowNet runs on the GPU only (no CPU fallback)")
        if log:
            raise NotImplementedError("attention-score logging is analysis tooling outside the forward path")
        with torch.no_grad():
            if x.size(1) != self.num_bins:                     # DSEC double-chunk input: last block of chunk 1 + chunk 2
                c1, c2 = x[:, :self.num_bins], x[:, self.num_bins:]
                if self.norm_input:
                    c1, c2 = self.normalize(c1), self.normalize(c2)
                parts = (c1.chunk(self.num_split, dim=1)[-1],) + tuple(c2.chunk(self.num_split, dim=1))
            else:
                parts = x.chunk(self.num_split, dim=1)
            x = torch.stack(list(parts), dim=0)                # (T,B,bins/T,H,W)
            H, W = x.shape[-2:]
            if H % 2 or W % 2:
                x = F.pad(x, (0, W % 2, 0, H % 2))
            flows, _ = self.sttmultires_unet(x)
            flow_list = [F.interpolate(f, scale_factor=(H / f.shape[2], W / f.shape[3])) for f in flows]
        return {"flow": flow_list, "attn": None, "spiking_rates": None}


class STTFlowNet_4en(STTFlowNet):
    num_en = 4
