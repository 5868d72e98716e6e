"""HIP execution engine of the SEW (spike-element-wise shortcut) family: `SpikingformerFlowNet`
(reference models/STSwinNet_SNN/Spiking_STSwinNet.py:254-311, Spiking_swin_transformer3D.py:115-162, 184-370, 720-950,
Spiking_modules.py:397-456, 571-603, 827-878).

Same patch embedding as the MS models (inherited): its output is a real-valued membrane, and the stream between blocks is that
membrane plus a SUM of spike tensors - real numbers in stage 0, small non-negative integers behind the first patch merging (whose
output is spikes).  The Linear layers that read the stream (q / k / v, fc1, the patch-merging reduction) and the projection behind the
real-valued attention output run on `sdf_dense_linear_fwd` (both operands as fp16 hi + lo planes on the 16-bit matrix pipe - exact to
22 bits for integers and reals alike); the first convolution of a res-block reads the integer stream of the last stage as BYTES through
the spike convolution kernels (whose fp16 A-operand expansion takes any byte value exactly, csrc/spike_mm.h `expand_spikes`);
everything that reads spikes (fc2, the second res-block convolution, the flow predictions) runs on the spike kernels, every BatchNorm
is folded into the neuron kernel that follows it.  The transposed convolutions of the decoders read a concatenation of a real-valued
prediction, spikes and a skip tensor (real-valued at the last level): they run as the dense 3x3 convolution (csrc/dense_conv_wres.hip,
chained over 96-channel record groups) of the zero-upsampled input with the flipped kernel - ConvTranspose2d(k 3, s 2, p 1, op 1) written
as a correlation.  No rocBLAS / MIOpen kernel is left in a SEW forward (profiles/r3q_sew_kernel_stats.txt; MIOpen served the transposed
convolutions with its naive fallback kernel: 27.6 of the forward's 35 ms), the score / bias / mask / .V core is the fused window-attention kernel (csrc/win_attn.hip, SEW mode); the window
partition is the row map the q / k / v neuron kernel gathers through (the projections run on un-partitioned rows), the window
reverse a row scatter through the same map (no materialised pad / roll / permute / crop).
No CPU fallback: everything here raises off-GPU."""
from __future__ import annotations

import os

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
        self.wq, self.wk, self.wv = (hip.pack_dense_linear_weight(getattr(a, f"linear_{n}").weight.detach().float().to(device)) for n in "qkv")
        self.bn = {n: bn_affine(getattr(a, f"bn_{n}").norm_layer, device) for n in "qkv"}
        self.sn = {n: _np(getattr(a, f"sn_{n}"), device) for n in "qkv"}
        self.scale = torch.full((a.num_heads,), float(a.scale), device=device)
        self.table = a.relative_position_bias_table.detach().float().to(device)
        self.index = a.relative_position_index.to(device)
        self.wp = hip.pack_dense_linear_weight(a.proj.weight.detach().float().to(device))
        self.bp = a.proj.bias.detach().float().contiguous().to(device)
        self.proj_bn, self.proj_sn = bn_affine(a.proj_bn.norm_layer, device), _np(a.proj_sn, device)
        self.w1 = hip.pack_dense_linear_weight(m.fc1.weight.detach().float().to(device))
        self.bn1, self.sn1 = bn_affine(m.bn1.norm_layer, device), _np(m.sn1, device)
        self.fc2 = _Lin(m.fc2, m.bn2.norm_layer, device, nsplit, digits=False)   # fc2 reads spikes: the spike GEMM (fp16 planes only)
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
                self.merges.append((hip.pack_dense_linear_weight(d.reduction.weight.detach().float().to(dev)), bn_affine(d.norm.norm_layer, dev),
                                    _np(d.sn, dev)))
        # The first res-block convolution reads the stream of the last stage as BYTES (`_stream_bytes`): that is exact only where the
        # stream is a sum of spike tensors - behind a patch merging (whose output is spikes) every block adds two spike tensors
        # (reference Spiking_swin_transformer3D.py:840-845).  Checked HERE, once, instead of by a host sync per forward: a model
        # without a patch merging would feed the real-valued patch embedding into the byte cast, and the int8 small-M kernel
        # (csrc/ms_smallm.hip) reads bytes >= 128 as negative.
        if len(unet.resblocks) > 0:
            if not self.merges:
                raise hip.SdfError("SEW family: a model with a single swin stage feeds the real-valued patch embedding into the res-blocks; "
                                   "the HIP engine reads that stream as sums of spikes (bytes) - no kernel is built for it")
            peak = 1 + 2 * len(self.stages[-1])
            if peak >= 128:
                raise hip.SdfError(f"SEW family: {len(self.stages[-1])} blocks in the last stage can sum to {peak} spikes per element; the "
                                   "res-block convolution reads the stream as int8 bytes (< 128)")
        self.unet_res = []
        for i, rb in enumerate(unet.resblocks):
            self.unet_res.append(_ResBlock(rb, dev, ns, U + f"resblocks.{i}."))  # conv2 reads spikes, conv1 the integer stream as bytes
        self.decoders = [(d.deconv[0].weight.detach().float().to(dev), bn_affine(d.norm_layer.norm_layer, dev), _np(d.sn, dev))
                         for d in unet.decoders]
        self._deconv_dense = {}
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
        if self.tape is not None:                     # (the byte copy is the tape's: not made when nothing records it)
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
            y = hip.dense_linear(x2, w)
            spk[n] = torch.empty((M, Cc), dtype=torch.uint8, device=x.device)
            hip.neuron_fwd(y, spk[n], Tq, 1, M // Tq * Cc, 0, 0, 0, M // Tq * Cc, blk.sn[n], rowmap=rowmap, rowlen=Cc,
                           alpha=blk.bn[n][0], beta=blk.bn[n][1], Cch=Cc, inner=1)
            self._rec(blk.name + f"attn.sn_{n}.spiking_neuron.", spk[n], "flat")
        mask = self._mask(D, H, W, ws, ss) if any(s > 0 for s in ss) else None
        z = hip.win_attn_sew(spk["q"], spk["k"], spk["v"], blk.scale, blk.bias(Tq * N1), mask, blk.nH, Tq, B_, N1)
        y = hip.dense_linear(z.view(M, Cc), blk.wp, blk.bp)
        s = self._sn_rows(y, Tq, blk.proj_sn, blk.proj_bn, blk.name + "attn.proj_sn.spiking_neuron.")
        return hip.rows_scatter(s, rowmap, B * D * H * W).view(B, D, H, W, Cc)     # window reverse (+ roll back, crop)

    def mlp(self, x, blk: _SewBlock):
        """MLP(x) over the true time axis D (reference :147-162): spikes, (B,D,H,W,C) fp32."""
        B, D, H, W, Cc = x.shape
        h = hip.dense_linear(x.view(-1, Cc), blk.w1).view(B, D, H, W, -1)
        s1 = self._neuron_bd(h, blk.sn1, bn=blk.bn1)
        self._rec(blk.name + "mlp.sn1.spiking_neuron.", s1, "BDHWC->TBHWC")
        y = torch.empty((B, D, H, W, Cc), dtype=torch.float32, device=x.device)
        hip.spike_gemm(s1, blk.fc2.Wp, y, B * D * H * W, Cc, blk.fc2.K, alpha=blk.fc2.alpha, beta=blk.fc2.beta)
        s2 = self._neuron_bd(y, blk.sn2, out_dtype=torch.float32)
        if self.tape is not None:
            self._rec(blk.name + "mlp.sn2.spiking_neuron.", s2.to(torch.uint8), "BDHWC->TBHWC")
        return s2

    def swin_block(self, x, s, i):
        blk = self.stages[s][i]
        x = self.attention(x, blk) + x                                            # SEW ADD (:840)
        return self.mlp(x, blk) + x                                               # (:845)

    def _tail_kwargs(self):
        return {}

    def _next_spikes(self, x, blk, sn):
        return None                                         # the SEW stream is not a membrane: every layer runs its own neuron

    def patch_merge(self, x, s, packed=None):
        """2x2 gather -> Linear -> BN -> SN (reference :914-934): spikes (B,D,H/2,W/2,2C) fp32."""
        wpk, bn, sn = self.merges[s] if packed is None else packed
        self._check_cl(x)
        B, D, H, W, Cc = x.shape
        rowmap, H2, W2, _ = self._merge_map(B, D, H, W)
        rows = B * H2 * W2
        xg = hip.rows_gather(x.view(-1, Cc), rowmap).view(D * rows, 4 * Cc)       # (t, b, h2, w2) rows of the 4C concat
        out = self._sn_rows(hip.dense_linear(xg, wpk), D, sn, bn, f"sttmultires_unet.encoders.swin3d.layers.{s}.downsample.sn.spiking_neuron.")
        out = out.view(D, B, H2, W2, -1)
        return out.permute(1, 0, 2, 3, 4).contiguous() if B > 1 else out.view(B, D, H2, W2, -1)

    def _stream_bytes(self, x):
        """The integer-valued SEW stream as the byte operand of the spike kernels.  Only the fp16-plane kernels (gemm_nsplit == 2) take
        any byte exactly (csrc/spike_mm.h expand_spikes: {n, 0x64} = fp16(1024 + n)); the bf16-plane expansion is for {0, 1} only, so the
        other weight modes are refused here instead of returning wrong sums; the int8 digit kernels (csrc/ms_smallm.hip) take
        bytes below 128.  That the stream IS a sum of at most 127 spikes is established at construction (`_init_stages`: a patch merging
        ahead of the res-blocks, the block count of the last stage); SDF_DEBUG_CHECKS=1 additionally verifies it on the host per call
        (a sync: off by default so that a SEW forward can be captured into a HIP graph)."""
        if self.nsplit != 2:
            raise hip.SdfError(f"SEW family: gemm_nsplit = {self.nsplit} is not built for the res-block convolution on the integer stream "
                               "(bf16 planes expand spikes {0, 1} only); use the default gemm_nsplit = 2")
        xu = x.to(torch.uint8)
        if hip.sw("SDF_DEBUG_CHECKS", "") == "1" and not (torch.equal(xu.float(), x) and int(xu.max()) < 128):
            raise hip.SdfError("the SEW stream in front of a res-block is not a sum of spike tensors (a model with a single swin "
                               "stage feeds the real-valued patch embedding here)")
        return xu

    def _sew_resblock(self, x, rb):
        """conv-BN-SN-conv-BN-SN + identity (reference Spiking_modules.py:852-878)."""
        B, D, h, w, Cc = x.shape
        # the stream behind a patch merging is a sum of spike tensors: exact bytes (at most one spike per block half and level)
        y = self._conv3x3(self._stream_bytes(x), rb.w1, rb.C)
        s1 = self._neuron_bd(y, rb.sn1, bn=rb.bn1)
        self._rec(rb.name + "sn1.spiking_neuron.", s1, "BDHWC->TBCHW")
        s2 = self._neuron_bd(self._conv3x3(s1, rb.w2, rb.C, bn=rb.bn2), rb.sn2, out_dtype=torch.float32)
        if self.tape is not None:
            self._rec(rb.name + "sn2.spiking_neuron.", s2.to(torch.uint8), "BDHWC->TBCHW")
        return s2 + x

    def _deconv_dense_fwd(self, i, parts, wdec):
        """ConvTranspose2d(k 3, s 2, p 1, output_padding 1) of cat(parts, channels) (reference Spiking_modules.py:449-456) on the dense
        convolution kernel: out[o] = sum_k u[o + 1 - k] w[k] with u the zero-upsampled input (u[2 i] = x[i]) is the 3x3 / pad 1
        correlation of u with the flipped kernel.  `parts`: channel-last (B,D,h,w,c) tensors in the reference's channel order, all of
        one size; physically each starts on a 16-channel record and the record count is brought to 6 k or 6 k + 1 (what
        `hip.dense_conv_slices` chains), the padding channels are zero in the input and in the weights."""
        B, D, h, w, _ = parts[0].shape
        imgs, cout = B * D, wdec.shape[1]
        cs = [p.shape[-1] for p in parts]
        recs = [-(-c // 16) for c in cs]
        total = sum(recs)
        while total % 6 > 1:
            total += 1
        if cout % 32 or hip.dense_conv_slices(total) is None or imgs * 4 * h * w * max(total * 64, cout * 4) >= 1 << 31:
            cat = torch.cat(parts, dim=-1)                                         # shapes outside the kernel's build: the library
            z = F.conv_transpose2d(cat.view(imgs, h, w, -1).permute(0, 3, 1, 2), wdec, None, stride=2, padding=1, output_padding=1)
            return z.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1).view(B, D, 2 * h, 2 * w, -1)
        key = (i, tuple(cs))
        if key not in self._deconv_dense:
            wphys = torch.zeros((total * 16, cout, 3, 3), dtype=torch.float32, device=wdec.device)
            c_ref = r0 = 0
            for c, r in zip(cs, recs):
                wphys[16 * r0:16 * r0 + c] = wdec[c_ref:c_ref + c]
                c_ref, r0 = c_ref + c, r0 + r
            wconv = wphys.flip(2, 3).permute(1, 0, 2, 3).contiguous()              # (Cout, Cin, 3, 3) of the equivalent correlation
            self._deconv_dense[key] = [((a, n), hip.pack_dense_conv_weight(wconv[:, 16 * a:16 * (a + n)]))
                                       for a, n in hip.dense_conv_slices(total)]
        # the zero-upsampled input goes straight into the activation planes (round 6: a zero-filled fp32 image of 16 x total channels at
        # twice the size, its strided fills and a packing pass stood here: 0.3 ms of a 4.3 ms forward)
        planes = torch.empty((imgs, total, 2 * h, 2 * w, 32), dtype=torch.float16, device=wdec.device)
        r0 = 0
        for p_, c, r in zip(parts, cs, recs):
            hip.pack_planes_zero_up2(p_.reshape(imgs, h, w, c).permute(0, 3, 1, 2), planes, r0)
            r0 += r
        if r0 < total:
            planes[:, r0:].zero_()                                                 # (the records that only round the count up to 6 k / 6 k + 1)
        z = hip.dense_conv3x3_wide(planes, self._deconv_dense[key], out_f32=True)
        return z.view(B, D, 2 * h, 2 * w, cout)

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
            wdec, bn, sn = self.decoders[i]
            z = self._deconv_dense_fwd(i, parts, wdec)
            sp = self._neuron_bd(z, sn, bn=bn)                                    # u8 spikes
            self._rec(f"sttmultires_unet.decoders.{i}.sn.spiking_neuron.", sp, "BDHWC->TBCHW")
            pw, pb, nout = self.preds[i]
            po = torch.empty((B * D * 4 * h * w, 32), dtype=torch.float32, device=y.device)
            hip.spike_gemm(sp, pw, po, po.shape[0], 32, sp.shape[-1], bias=pb)
            preds.append(po.view(B, D, 2 * h, 2 * w, 32))
            y = sp.float()
        return [p[..., :self.preds[0][2]] for p in preds]
