"""HIP execution engine of the SEW (spike-element-wise shortcut) family: `SpikingformerFlowNet`
(reference models/STSwinNet_SNN/Spiking_STSwinNet.py:254-311, Spiking_swin_transformer3D.py:115-162, 184-370, 720-950,
Spiking_modules.py:397-456, 571-603, 827-878).

Same patch embedding as the MS models (inherited).  Behind it the stream between blocks is a SUM of spike tensors, so the
layers that read it see small non-negative numbers, not spikes: those products (q/k/v and fc1 projections, patch-merging
reduction, the first convolution of a res-block, the transposed convolutions) are dense fp32 library calls (rocBLAS /
MIOpen through torch - plain library GEMMs / convolutions), everything that reads spikes (fc2, the second res-block
convolution, the flow predictions) runs on the spike kernels, every BatchNorm is folded into the neuron kernel that
follows it, the score / bias / mask / .V core is the fused window-attention kernel (csrc/win_attn.hip, SEW mode); the window
partition is the row map the q / k / v neuron kernel gathers through (the projections run on un-partitioned rows), the window
reverse a row scatter through the same map (no materialised pad / roll / permute / crop).
No CPU fallback: everything here raises off-GPU."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import hip
from .engine import MSFlowEngine, _Lin, _ResBlock, _conv_planes, _np, _pad32, bn_affine
from .STSwinNet.swin_transformer3D_v2 import compute_mask
from .STSwinNet_SNN.Spiking_swin_transformer3D import get_window_size


class _SewBlock:
    def __init__(self, blk, device, nsplit, name):
        a, m = blk.attn, blk.mlp
        self.name, self.nH, self.window_size, self.shift_size = name, a.num_heads, blk.window_size, blk.shift_size
        self.wq, self.wk, self.wv = (getattr(a, f"linear_{n}").weight.detach().float().t().contiguous().to(device) for n in "qkv")
        self.bn = {n: bn_affine(getattr(a, f"bn_{n}").norm_layer, device) for n in "qkv"}
        self.sn = {n: _np(getattr(a, f"sn_{n}"), device) for n in "qkv"}
        self.scale = torch.full((a.num_heads,), float(a.scale), device=device)
        self.table = a.relative_position_bias_table.detach().float().to(device)
        self.index = a.relative_position_index.to(device)
        self.wp_t = a.proj.weight.detach().float().t().contiguous().to(device)
        self.bp = a.proj.bias.detach().float().contiguous().to(device)
        self.proj_bn, self.proj_sn = bn_affine(a.proj_bn.norm_layer, device), _np(a.proj_sn, device)
        self.w1_t = m.fc1.weight.detach().float().t().contiguous().to(device)
        self.bn1, self.sn1 = bn_affine(m.bn1.norm_layer, device), _np(m.sn1, device)
        self.fc2 = _Lin(m.fc2, m.bn2.norm_layer, device, nsplit)                 # fc2 reads spikes: the spike GEMM
        self.sn2 = _np(m.sn2, device)
        self._bias = {}

    def bias(self, N):
        if N not in self._bias:
            idx = self.index[:N, :N].reshape(-1)
            self._bias[N] = self.table[idx].reshape(N, N, -1).permute(2, 0, 1).contiguous()
        return self._bias[N]


class SEWFlowEngine(MSFlowEngine):
    def _init_stages(self, model, unet, sw, dev, ns, U):
        self.stages, self.merges = [], []
        for li, layer in enumerate(sw.layers):
            self.stages.append([_SewBlock(b, dev, ns, U + f"encoders.swin3d.layers.{li}.swin_blocks.{bi}.")
                                for bi, b in enumerate(layer.swin_blocks)])
            if layer.downsample is not None:
                d = layer.downsample
                self.merges.append((d.reduction.weight.detach().float().t().contiguous().to(dev), bn_affine(d.norm.norm_layer, dev),
                                    _np(d.sn, dev)))
        self.unet_res = []
        for i, rb in enumerate(unet.resblocks):
            r = _ResBlock(rb, dev, ns, U + f"resblocks.{i}.")                    # conv2 reads spikes: planes; conv1 reads the stream
            r.w1_dense = rb.conv1[0].weight.detach().float().contiguous(memory_format=torch.channels_last).to(dev)
            self.unet_res.append(r)
        self.decoders = [(d.deconv[0].weight.detach().float().to(dev), bn_affine(d.norm_layer.norm_layer, dev), _np(d.sn, dev))
                         for d in unet.decoders]
        self.preds = []
        for p in unet.preds:
            w2 = p.conv[0].weight.detach().float().reshape(p.conv[0].weight.shape[0], -1)
            wp = torch.zeros((32, w2.shape[1]), dtype=torch.float32, device=dev)
            wp[:w2.shape[0]] = w2
            bp = torch.zeros(32, dtype=torch.float32, device=dev)
            bp[:w2.shape[0]] = p.conv[0].bias.detach().float()
            self.preds.append((hip.split_weight(wp, ns), bp, w2.shape[0]))
        self._masks = {}

    # ------------------------------------------------------------------ helpers
    def _mask(self, D, H, W, ws, ss):
        key = (D, H, W, ws, ss)
        if key not in self._masks:
            Dp, Hp, Wp = (-(-D // ws[0])) * ws[0], (-(-H // ws[1])) * ws[1], (-(-W // ws[2])) * ws[2]
            self._masks[key] = compute_mask(Dp, Hp, Wp, ws, ss, self.device).contiguous()
        return self._masks[key]

    def _sn_rows(self, y, T, p, bn, name, out_dtype=torch.float32):
        """Neuron over the leading T blocks of a (T * rows, C) buffer, BN folded in; returns (T * rows, C) spikes."""
        M, Cc = y.shape
        out = torch.empty((M, Cc), dtype=out_dtype, device=y.device)
        n = M // T * Cc
        hip.neuron_fwd(y, out, T, 1, n, 0, n, 0, n, p, alpha=bn[0], beta=bn[1], Cch=Cc, inner=1)
        self._rec(name, out.to(torch.uint8) if out_dtype != torch.uint8 else out, "flat")
        return out

    # ------------------------------------------------------------------ stages
    def attention(self, x, blk: _SewBlock):
        """SSA(x) (reference Spiking_swin_transformer3D.py:781-821 around :300-370): spikes, (B,D,H,W,C) fp32."""
        self._check_cl(x)
        B, D, H, W, Cc = x.shape
        ws, ss = get_window_size((D, H, W), blk.window_size, blk.shift_size)
        if tuple(ws) != tuple(blk.window_size):
            # the reference clamps the window and then cannot add its (N, N) relative-position table, indexed for the nominal
            # window, to the smaller scores (Spiking_swin_transformer3D.py:345-350, RuntimeError); refused here as well
            raise hip.SdfError(f"feature map {(D, H, W)} is smaller than the window {tuple(blk.window_size)}: the relative position bias of "
                               "Spiking_BN_WindowAttention3D is defined for the nominal window only")
        rowmap, B_ = self._slice_map(B, D, H, W, ws, ss)
        Tq, N1 = ws[0], ws[1] * ws[2]
        M = Tq * B_ * N1
        # q / k / v: the projection is row-wise, so it runs on the un-partitioned rows; pad + roll + window_partition_v2 (and its raw
        # (T', B_, ...) view) are the row map the neuron kernel gathers through - nothing is materialised.  A padding token reads
        # zeros: BN(Linear(0)) = beta, as in the reference (the projections carry no bias)
        x2 = x.view(-1, Cc)
        spk = {}
        for n, w in (("q", blk.wq), ("k", blk.wk), ("v", blk.wv)):
            y = torch.mm(x2, w)
            spk[n] = torch.empty((M, Cc), dtype=torch.uint8, device=x.device)
            hip.neuron_fwd(y, spk[n], Tq, 1, M // Tq * Cc, 0, 0, 0, M // Tq * Cc, blk.sn[n], rowmap=rowmap, rowlen=Cc,
                           alpha=blk.bn[n][0], beta=blk.bn[n][1], Cch=Cc, inner=1)
            self._rec(blk.name + f"attn.sn_{n}.spiking_neuron.", spk[n], "flat")
        mask = self._mask(D, H, W, ws, ss) if any(s > 0 for s in ss) else None
        z = hip.win_attn_sew(spk["q"], spk["k"], spk["v"], blk.scale, blk.bias(Tq * N1), mask, blk.nH, Tq, B_, N1)
        y = torch.addmm(blk.bp, z.view(M, Cc), blk.wp_t)
        s = self._sn_rows(y, Tq, blk.proj_sn, blk.proj_bn, blk.name + "attn.proj_sn.spiking_neuron.")
        return hip.rows_scatter(s, rowmap, B * D * H * W).view(B, D, H, W, Cc)     # window reverse (+ roll back, crop)

    def mlp(self, x, blk: _SewBlock):
        """MLP(x) over the true time axis D (reference :147-162): spikes, (B,D,H,W,C) fp32."""
        B, D, H, W, Cc = x.shape
        h = torch.mm(x.view(-1, Cc), blk.w1_t).view(B, D, H, W, -1)
        s1 = self._neuron_bd(h, blk.sn1, bn=blk.bn1)
        self._rec(blk.name + "mlp.sn1.spiking_neuron.", s1, "BDHWC->TBHWC")
        y = torch.empty((B, D, H, W, Cc), dtype=torch.float32, device=x.device)
        hip.spike_gemm(s1, blk.fc2.Wp, y, B * D * H * W, Cc, blk.fc2.K, alpha=blk.fc2.alpha, beta=blk.fc2.beta)
        s2 = self._neuron_bd(y, blk.sn2, out_dtype=torch.float32)
        self._rec(blk.name + "mlp.sn2.spiking_neuron.", s2.to(torch.uint8), "BDHWC->TBHWC")
        return s2

    def swin_block(self, x, s, i):
        blk = self.stages[s][i]
        x = self.attention(x, blk) + x                                            # SEW ADD (:840)
        return self.mlp(x, blk) + x                                               # (:845)

    def patch_merge(self, x, s, packed=None):
        """2x2 gather -> Linear -> BN -> SN (reference :914-934): spikes (B,D,H/2,W/2,2C) fp32."""
        w_t, bn, sn = self.merges[s] if packed is None else packed
        self._check_cl(x)
        B, D, H, W, Cc = x.shape
        rowmap, H2, W2, _ = self._merge_map(B, D, H, W)
        rows = B * H2 * W2
        xg = hip.rows_gather(x.view(-1, Cc), rowmap).view(D * rows, 4 * Cc)       # (t, b, h2, w2) rows of the 4C concat
        out = self._sn_rows(torch.mm(xg, w_t), D, sn, bn, f"sttmultires_unet.encoders.swin3d.layers.{s}.downsample.sn.spiking_neuron.")
        out = out.view(D, B, H2, W2, -1)
        return out.permute(1, 0, 2, 3, 4).contiguous() if B > 1 else out.view(B, D, H2, W2, -1)

    def _sew_resblock(self, x, rb):
        """conv-BN-SN-conv-BN-SN + identity (reference Spiking_modules.py:852-878)."""
        B, D, h, w, Cc = x.shape
        y = F.conv2d(x.view(B * D, h, w, Cc).permute(0, 3, 1, 2), rb.w1_dense, None, 1, 1)
        y = y.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).view(B, D, h, w, -1)
        s1 = self._neuron_bd(y, rb.sn1, bn=rb.bn1)
        self._rec(rb.name + "sn1.spiking_neuron.", s1, "BDHWC->TBCHW")
        s2 = self._neuron_bd(self._conv3x3(s1, rb.w2, rb.C, bn=rb.bn2), rb.sn2, out_dtype=torch.float32)
        self._rec(rb.name + "sn2.spiking_neuron.", s2.to(torch.uint8), "BDHWC->TBCHW")
        return s2 + x

    def unet_tail(self, feats, out_size=None):
        """SEW res-blocks + decoders (ConvT -> BN -> SN) + plain 1x1 predictions (reference Spiking_STSwinNet.py:161-182)."""
        self._flows = [None] * len(feats)                                          # (flow read-out: sdf_flow_out_fwd per scale in forward())
        y = feats[-1]
        for rb in self.unet_res:
            y = self._sew_resblock(y, rb)
        preds, E = [], len(feats)
        for i in range(E):
            skip = feats[E - 1 - i]
            B, D, h, w, _ = skip.shape
            parts = ([preds[-1][..., :self.preds[i - 1][2]]] if i > 0 else []) + [y, skip]      # [prediction | y | skip] (:168-172)
            parts = [F.pad(p, (0, 0, (w - p.shape[3]) // 2, w - p.shape[3] - (w - p.shape[3]) // 2,
                               (h - p.shape[2]) // 2, h - p.shape[2] - (h - p.shape[2]) // 2)) for p in parts]
            cat = torch.cat(parts, dim=-1)
            wdec, bn, sn = self.decoders[i]
            z = F.conv_transpose2d(cat.view(B * D, h, w, -1).permute(0, 3, 1, 2), wdec, None, stride=2, padding=1, output_padding=1)
            z = z.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).view(B, D, 2 * h, 2 * w, -1)
            sp = self._neuron_bd(z, sn, bn=bn)                                    # u8 spikes
            self._rec(f"sttmultires_unet.decoders.{i}.sn.spiking_neuron.", sp, "BDHWC->TBCHW")
            pw, pb, nout = self.preds[i]
            po = torch.empty((B * D * 4 * h * w, 32), dtype=torch.float32, device=y.device)
            hip.spike_gemm(sp, pw, po, po.shape[0], 32, sp.shape[-1], bias=pb)
            preds.append(po.view(B, D, 2 * h, 2 * w, 32))
            y = sp.float()
        return [p[..., :self.preds[0][2]] for p in preds]
