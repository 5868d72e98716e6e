"""HIP execution engine of the MS (membrane-shortcut) SDformerFlow family.

Turns the module tree (state_dict holder) into a flat schedule of C-ABI kernel launches:

  * weights of every spike-fed Linear are pre-split into bf16 planes (fp32-grade on the bf16 MFMA path)
  * eval-BatchNorm becomes (alpha, beta) vectors applied as one fmaf inside the producing / consuming kernel
  * pad + roll + window_partition_v2 (+ the raw .view) and window_reverse + roll + crop become one int32
    row map per (shape, shift), used as a gather by the neuron kernel and as a scatter by the GEMM epilogue
  * the residual adds of the block happen in the GEMM epilogues, in place, on the (B,D,H,W,C) activation
  * spikes travel as 1 byte between kernels (fp32 only where a MIOpen convolution consumes them)

Convolutions (patch embedding, U-Net tail: SURVEY.md 8f rows 1-2) are stock ATen/MIOpen calls for now.
Reference schedule being replaced: models/STSwinNet_SNN/Spiking_STSwinNet.py:161-182,278-305 and the
call tree of SURVEY.md 3.2.  No CPU fallback: everything here raises off-GPU.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import hip
from .STSwinNet_SNN.Spiking_swin_transformer3D import get_window_size, merge_row_map, window_slice_map

BN_EPS = 1e-5


def bn_affine(bn, device):
    """(alpha, beta) of an eval BatchNorm, computed on the host exactly like ATen's CPU kernel:
    alpha = w * rsqrt(var + eps) (fp32), beta = fma(-mean, alpha, bias)."""
    w, b = bn.weight.detach().float().cpu(), bn.bias.detach().float().cpu()
    m, v = bn.running_mean.detach().float().cpu(), bn.running_var.detach().float().cpu()
    alpha = w * torch.rsqrt(v + BN_EPS)
    beta = (b.double() - m.double() * alpha.double()).float()
    return alpha.contiguous().to(device), beta.contiguous().to(device)


class _Lin:
    """A Linear fed by spikes: bf16 weight planes + optional bias + the BN that follows it."""

    def __init__(self, linear, bn, device, nsplit):
        w = linear.weight.detach().float().to(device).contiguous()
        self.N, self.K = w.shape
        self.Wp = hip.split_weight(w, nsplit)
        self.bias = None if linear.bias is None else linear.bias.detach().float().to(device).contiguous()
        self.alpha, self.beta = bn_affine(bn, device) if bn is not None else (None, None)


def _np(sn_module, device):
    """NeuronParams of a Spiking_neuron wrapper, PSN weights moved to the device."""
    n = sn_module.spiking_neuron
    if n.kind == "psn":
        return hip.NeuronParams("psn", psn_w=n.weight.detach().float().to(device).contiguous(),
                                psn_b=n.bias.detach().float().reshape(-1).to(device).contiguous())
    return hip.NeuronParams(n.kind, n.tau, n.v_threshold, n.v_reset)


class _Block:
    def __init__(self, blk, device, nsplit):
        a = blk.attn
        self.nH, self.window_size, self.shift_size = a.num_heads, blk.window_size, blk.shift_size
        self.q = _Lin(a.linear_q, a.bn_q.norm_layer, device, nsplit)
        self.k = _Lin(a.linear_k, a.bn_k.norm_layer, device, nsplit)
        self.p = _Lin(a.proj, a.proj_bn.norm_layer, device, nsplit)
        self.pe = a.positional_encoding.detach().float().to(device).contiguous()
        self.sn_proj, self.sn_q, self.sn_k, self.sn2_q = (_np(m, device) for m in (a.proj_sn, a.sn_q, a.sn_k, a.sn2_q))
        m = blk.mlp
        self.fc1 = _Lin(m.fc1, m.bn1.norm_layer, device, nsplit)
        self.fc2 = _Lin(m.fc2, m.bn2.norm_layer, device, nsplit)
        self.sn1, self.sn2 = _np(m.sn1, device), _np(m.sn2, device)


class _ResBlock:
    def __init__(self, rb, device):
        self.w1, self.w2 = rb.conv1[0].weight.detach(), rb.conv2[0].weight.detach()
        self.bn1, self.bn2 = bn_affine(rb.norm1.norm_layer, device), bn_affine(rb.norm2.norm_layer, device)
        self.sn1, self.sn2 = _np(rb.sn1, device), _np(rb.sn2, device)


class MSFlowEngine:
    def __init__(self, model):
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise hip.SdfError("the SDformerFlow forward runs on the MI355X HIP engine only: call model.to('cuda') "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
        hip.lib()
        dev, ns = self.device, model.gemm_nsplit
        unet = model.sttmultires_unet
        sw = unet.encoders.swin3d
        pe = sw.patch_embed
        self.num_bins, self.num_steps = pe.num_bins, pe.num_steps
        self.head_w = pe.head.conv[0].weight.detach()
        self.head_bn, self.head_sn = bn_affine(pe.head.norm_layer.norm_layer, dev), _np(pe.head.sn, dev)
        self.conv_w, self.conv_bn = pe.conv.conv[0].weight.detach(), bn_affine(pe.conv.norm_layer.norm_layer, dev)
        self.pe_res = [_ResBlock(rb, dev) for rb in pe.residual_encoding.resblocks]
        self.proj_res_w, self.proj_w = pe.proj.conv_res.weight.detach(), pe.proj.conv.weight.detach()
        self.proj_bn, self.proj_sn = bn_affine(pe.proj.norm_layer, dev), _np(pe.proj.sn, dev)
        self.stages, self.merges = [], []
        for layer in sw.layers:
            self.stages.append([_Block(b, dev, ns) for b in layer.swin_blocks])
            if layer.downsample is not None:
                d = layer.downsample
                self.merges.append((_Lin(d.reduction, d.norm.norm_layer, dev, ns), _np(d.sn, dev)))
        self.unet_res = [_ResBlock(rb, dev) for rb in unet.resblocks]
        self.decoders = [(d.deconv[0].weight.detach(), bn_affine(d.norm_layer.norm_layer, dev), _np(d.sn, dev),
                          d.deconv[0].kernel_size[0]) for d in unet.decoders]
        self.preds = [(p.conv[0].weight.detach(), p.conv[0].bias.detach(), _np(p.sn, dev)) for p in unet.preds]
        self._maps = {}

    # ------------------------------------------------------------------ helpers
    def _slice_map(self, B, D, H, W, ws, ss):
        key = ("win", B, D, H, W, ws, ss)
        if key not in self._maps:
            src, B_ = window_slice_map(B, D, H, W, ws, ss)
            self._maps[key] = (torch.from_numpy(src.reshape(-1)).to(self.device), B_)
        return self._maps[key]

    def _merge_map(self, B, D, H, W):
        key = ("merge", B, D, H, W)
        if key not in self._maps:
            src, H2, W2 = merge_row_map(B, D, H, W)
            out_map = None
            if B > 1:       # GEMM rows are (t,b,h2,w2); the activation is laid out (b,t,h2,w2)
                t = torch.arange(D).view(D, 1, 1)
                b = torch.arange(B).view(1, B, 1)
                p = torch.arange(H2 * W2).view(1, 1, -1)
                out_map = ((b * D + t) * (H2 * W2) + p).reshape(-1).to(torch.int32).to(self.device)
            self._maps[key] = (torch.from_numpy(src.reshape(-1)).to(self.device), H2, W2, out_map)
        return self._maps[key]

    def _neuron_nchw(self, y, T, p, bn=None, out_dtype=torch.float32):
        """Neuron over the leading T of a (T*B, C, h, w) conv output, BN fused when given."""
        TB, Cc, h, w = y.shape
        n = (TB // T) * Cc * h * w
        out = torch.empty(y.shape, dtype=out_dtype, device=y.device)
        a, b = bn if bn is not None else (None, None)
        hip.neuron_fwd(y, out, T, 1, n, 0, n, 0, n, p, alpha=a, beta=b, Cch=Cc, inner=h * w)
        return out

    def _resblock(self, m, T, rb):
        """MS_ResBlock on a (T*B, C, h, w) membrane (reference Spiking_modules.py:906-933)."""
        Cc, hw = m.shape[1], m.shape[2] * m.shape[3]
        s = self._neuron_nchw(m, T, rb.sn1)
        y = F.conv2d(s, rb.w1, None, 1, 1)
        s = self._neuron_nchw(y, T, rb.sn2, rb.bn1)
        y = F.conv2d(s, rb.w2, None, 1, 1)
        return hip.affine_resid(y, rb.bn2[0], rb.bn2[1], Cc, hw, resid=m)

    # ------------------------------------------------------------------ stages (each usable stand-alone in tests)
    def patch_embed(self, x):
        """(B,bins,2,H,W) -> membrane (T,B,C,H/4,W/4) (reference Spiking_modules.py:1770-1790)."""
        if x.size(1) > self.num_bins:
            x = x[:, :self.num_bins]
        B, T = x.shape[0], self.num_steps
        num_ch = self.num_bins * 2 // T
        ev = x.permute(0, 2, 3, 4, 1)
        xr = torch.stack([ev[:, i % 2, :, :, (i // 2) * T:(i // 2 + 1) * T] for i in range(num_ch)], 1)
        xr = xr.permute(4, 0, 1, 2, 3).contiguous().flatten(0, 1)                        # (T*B, num_ch, H, W)
        y = F.conv2d(xr, self.head_w, None, 1, 1)
        s = self._neuron_nchw(y, T, self.head_sn, self.head_bn)
        y = F.conv2d(s, self.conv_w, None, 2, 1)
        m = hip.affine_resid(y, self.conv_bn[0], self.conv_bn[1], y.shape[1], y.shape[2] * y.shape[3])
        for rb in self.pe_res:
            m = self._resblock(m, T, rb)
        res = F.conv2d(m, self.proj_res_w, None, 2)
        s = self._neuron_nchw(m, T, self.proj_sn)
        z = F.conv2d(s, self.proj_w, None, 2, 1)
        out = hip.affine_resid(z, self.proj_bn[0], self.proj_bn[1], z.shape[1], z.shape[2] * z.shape[3], resid=res)
        return out.view(T, B, *out.shape[1:])

    def attention(self, x, blk: _Block):
        """x (B,D,H,W,C) += SSA(x), in place (reference Spiking_swin_transformer3D.py:781-821, 661-717, :840)."""
        B, D, H, W, Cc = x.shape
        ws, ss = get_window_size((D, H, W), blk.window_size, blk.shift_size)
        rowmap, B_ = self._slice_map(B, D, H, W, ws, ss)
        Tq, N1 = ws[0], ws[1] * ws[2]
        rows = B_ * N1
        n, M = rows * Cc, Tq * rows
        dev = x.device
        xs = torch.empty((M, Cc), dtype=torch.uint8, device=dev)
        hip.neuron_fwd(x, xs, Tq, 1, n, 0, 0, 0, n, blk.sn_proj, rowmap=rowmap, rowlen=Cc)
        # q = SN(BN(xs Wq^T)), k = SN(BN(xs Wk^T) + PE): GEMM with the neuron fused into its epilogue (u8 out)
        q = torch.empty((M, Cc), dtype=torch.uint8, device=dev)
        hip.spike_gemm_sn(xs, blk.q.Wp, q, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_q, alpha=blk.q.alpha, beta=blk.q.beta)
        k = torch.empty((M, Cc), dtype=torch.uint8, device=dev)
        hip.spike_gemm_sn(xs, blk.k.Wp, k, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_k, alpha=blk.k.alpha, beta=blk.k.beta,
                          add=blk.pe, add_prows=N1)
        hip.qk_gate(q, k, xs, Tq, rows, Cc, blk.sn2_q)                                  # E overwrites xs
        hip.spike_gemm(xs, blk.p.Wp, x, M, Cc, Cc, bias=blk.p.bias, alpha=blk.p.alpha, beta=blk.p.beta, resid=x,
                       out_rowmap=rowmap, zg=(blk.nH, Tq, B_, N1))
        return x

    def mlp(self, x, blk: _Block):
        """x (B,D,H,W,C) += MLP(x) over the true time axis D, in place (reference :164-181, :845)."""
        B, D, H, W, Cc = x.shape
        ntok, hw = B * D * H * W, H * W
        Ch = blk.fc1.N
        dev = x.device
        s1 = torch.empty((ntok, Cc), dtype=torch.uint8, device=dev)
        hip.neuron_fwd(x, s1, D, B, hw * Cc, D * hw * Cc, hw * Cc, D * hw * Cc, hw * Cc, blk.sn1)
        # s2 = SN(BN(s1 W1^T)) over the D time steps of every (b, h, w): fused GEMM + neuron, hidden never in fp32
        s2 = torch.empty((ntok, Ch), dtype=torch.uint8, device=dev)
        hip.spike_gemm_sn(s1, blk.fc1.Wp, s2, Ch, Cc, D, B * hw, hw, D * hw, hw, blk.sn2, alpha=blk.fc1.alpha,
                          beta=blk.fc1.beta)
        hip.spike_gemm(s2, blk.fc2.Wp, x, ntok, Cc, Ch, alpha=blk.fc2.alpha, beta=blk.fc2.beta, resid=x)
        return x

    def swin_block(self, x, s, i):
        blk = self.stages[s][i]
        return self.mlp(self.attention(x, blk), blk)

    def patch_merge(self, x, s):
        """(B,D,H,W,C) -> (B,D,H/2,W/2,2C) (reference :952-974)."""
        lin, sn = self.merges[s]
        B, D, H, W, Cc = x.shape
        rowmap, H2, W2, out_map = self._merge_map(B, D, H, W)
        rows = B * H2 * W2
        sp = torch.empty((D * rows, 4 * Cc), dtype=torch.uint8, device=x.device)
        hip.neuron_fwd(x, sp, D, 1, rows * 4 * Cc, 0, 0, 0, rows * 4 * Cc, sn, rowmap=rowmap, rowlen=Cc)
        out = torch.empty((B, D, H2, W2, lin.N), dtype=torch.float32, device=x.device)
        hip.spike_gemm(sp, lin.Wp, out, D * rows, lin.N, 4 * Cc, alpha=lin.alpha, beta=lin.beta, out_rowmap=out_map)
        return out

    def encoder(self, x):
        """-> per-stage features (D,B,C,h,w) (reference :1223-1246 + Spiking_STSwinNet.py:77-85)."""
        m = self.patch_embed(x)                                         # (T,B,C,h,w)
        y = m.permute(1, 0, 3, 4, 2).contiguous()                       # (B,D,h,w,C)
        feats = []
        for s, blocks in enumerate(self.stages):
            for i in range(len(blocks)):
                y = self.swin_block(y, s, i)
            feats.append(y.permute(1, 0, 4, 2, 3).contiguous())
            if s < len(self.merges):
                y = self.patch_merge(y, s)
        return feats

    @staticmethod
    def _skip_cat(x1, x2):
        dY, dX = x2.shape[-2] - x1.shape[-2], x2.shape[-1] - x1.shape[-1]
        if dY or dX:
            x1 = F.pad(x1, (dX // 2, dX - dX // 2, dY // 2, dY - dY // 2))
        return torch.cat([x1, x2], dim=2)

    def unet_tail(self, feats):
        """res-blocks + decoders + per-scale predictions (reference Spiking_STSwinNet.py:161-182)."""
        T, B = feats[-1].shape[:2]
        y = feats[-1].flatten(0, 1)
        for rb in self.unet_res:
            y = self._resblock(y, T, rb)
        y = y.view(T, B, *y.shape[1:])
        preds, E = [], len(feats)
        for i in range(E):
            y = self._skip_cat(y, feats[E - 1 - i])
            if i > 0:
                y = self._skip_cat(preds[-1], y)
            w, bn, sn, k = self.decoders[i]
            s = self._neuron_nchw(y.flatten(0, 1).contiguous(), T, sn)
            z = F.conv_transpose2d(s, w, None, stride=2, padding=k // 2, output_padding=1)
            z = hip.affine_resid(z, bn[0], bn[1], z.shape[1], z.shape[2] * z.shape[3])
            pw, pb, psn = self.preds[i]
            sp = self._neuron_nchw(z, T, psn)
            p = F.conv2d(sp, pw, pb)
            y = z.view(T, B, *z.shape[1:])
            preds.append(p.view(T, B, *p.shape[1:]))
        return preds

    def forward(self, x):
        """(B,bins,2,H,W) fp32 on the GPU -> list of E flow maps (B,2,H,W) (reference :278-305)."""
        if not x.is_cuda:
            raise hip.SdfError("input must be a GPU tensor (no CPU fallback)")
        x = x.float().contiguous()
        H, W = x.shape[-2:]
        preds = self.unet_tail(self.encoder(x))
        flows = []
        for p in preds:
            f = p.sum(0)
            flows.append(F.interpolate(f, scale_factor=(H / f.shape[-2], W / f.shape[-1])))
        return flows
