"""HIP execution engine of the MS (membrane-shortcut) SDformerFlow family.

Turns the module tree (state_dict holder) into a flat schedule of C-ABI kernel launches:

  * weights of every spike-fed Linear are pre-split into bf16 planes (fp32-grade on the bf16 MFMA path)
  * eval-BatchNorm becomes (alpha, beta) vectors applied as one fmaf inside the producing / consuming kernel
  * pad + roll + window_partition_v2 (+ the raw .view) and window_reverse + roll + crop become one int32
    row map per (shape, shift), used as a gather by the neuron kernel and as a scatter by the GEMM epilogue
  * the residual adds of the block happen in the GEMM epilogues, in place, on the (B,D,H,W,C) activation
  * spikes travel as 1 byte between kernels
  * the convolutions either side of the swin stages (patch embedding, U-Net tail: SURVEY.md 8f rows 1-2) are implicit-GEMM
    spike convolutions with the BN / residual / neuron epilogues fused; the large 96-channel 3x3 launches take int8 digit
    planes and the weight-resident kernel (csrc/spike_conv_wres.hip); no library kernel is left in the SNN forward
  * `forward(x, replicas=True)`: the B samples are B independent batch-1 forwards in one launch sequence (round 6; `_rb`, `_slice_map`)
  * `tape` (tests only) keeps every neuron layer's spikes for the spike-forced oracle replay (tests/replay.py)

Reference schedule being replaced: models/STSwinNet_SNN/Spiking_STSwinNet.py:161-182,278-305 and the
call tree of SURVEY.md 3.2.  No CPU fallback: everything here raises off-GPU.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import hip
from .STSwinNet_SNN.Spiking_swin_transformer3D import get_window_size, merge_row_map, window_slice_map

BN_EPS = 1e-5
# Narrowest stage that runs on int8 digit planes with the neurons in the producing kernels' epilogues: from C = 96 (swin stage 0) on
# since round 5 - stages 0 - 1 (C <= 192) on the weight-resident row-loop kernels of csrc/ms_res.hip, stages 2 - 3 on csrc/ms_wide.hip
# (round 4 had the wide kernels from C = 192 and qk_front / spike_gemm / ms_mlp_fused on the fp16 planes at stage 0).
_WIDE_MINC = 96


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

    def __init__(self, linear, bn, device, nsplit, digits=True):
        w = linear.weight.detach().float().to(device).contiguous()
        self.N, self.K = w.shape
        self.Wp = hip.split_weight(w, nsplit)
        # wide layers (swin stages 1 - 3) also carry int8 digit planes: what csrc/ms_wide.hip multiplies by (default plane mode only)
        # (`digits=False`: callers whose layer never reaches a digit kernel - the SEW family's fc2 - skip the extra 3 bytes per weight)
        self.digits = hip.split_weight_i8x3(w) if digits and nsplit == 2 and self.K >= _WIDE_MINC and self.K % 32 == 0 and self.N % 32 == 0 else None
        self.bias = None if linear.bias is None else linear.bias.detach().float().to(device).contiguous()
        self.alpha, self.beta = bn_affine(bn, device) if bn is not None else (None, None)


def _np(sn_module, device):
    """NeuronParams of a Spiking_neuron wrapper, PSN weights moved to the device."""
    n = sn_module.spiking_neuron
    if n.kind not in hip.KIND:
        raise hip.SdfError(f"neuron {type(n).__name__}: no fused kernel (lif / if / psn / plif / SLTTlif have one); it only runs as a "
                           "stand-alone module (reference Spiking_modules.py:84-92)")
    if n.kind == "psn":
        return hip.NeuronParams("psn", psn_w=n.weight.detach().float().to(device).contiguous(),
                                psn_b=n.bias.detach().float().reshape(-1).to(device).contiguous())
    return hip.NeuronParams(n.kind, n.tau, n.v_threshold, n.v_reset)


def attention_score(e, sn, nH, Tq, B_, N1):
    """`attn = self.attn_sn(x)` of `Spiking_QK_WindowAttention3D.forward` (reference Spiking_swin_transformer3D.py:709-711) from the
    gated spikes e (T', B_*N1, C) u8 of one block: the head scramble Z[t,b,n,g,d] = E_flat[((((b nH + g) T' + t) N1 + n) hd + d] as a
    re-layout (diagnostic path: the score is dead on the forward path, :715-716), then the neuron kernel over T'.
    -> (T', B_, N1, C) fp32 spikes."""
    Cc = e.shape[-1]
    z = e.reshape(B_, nH, Tq, N1, Cc // nH).permute(2, 0, 3, 1, 4).reshape(Tq, B_ * N1 * Cc).float().contiguous()
    out = torch.empty_like(z)
    n = z.shape[1]
    hip.neuron_fwd(z, out, Tq, 1, n, 0, n, 0, n, sn)
    return out.view(Tq, B_, N1, Cc)


class _Block:
    def __init__(self, blk, device, nsplit, name=""):
        self.name = name
        a = blk.attn
        self.attn_sn = _np(a.attn_sn, device)
        self.nH, self.window_size, self.shift_size = a.num_heads, blk.window_size, blk.shift_size
        self.q = _Lin(a.linear_q, a.bn_q.norm_layer, device, nsplit)
        self.k = _Lin(a.linear_k, a.bn_k.norm_layer, device, nsplit)
        self.p = _Lin(a.proj, a.proj_bn.norm_layer, device, nsplit)
        self.pe = a.positional_encoding.detach().float().to(device).contiguous()
        self.sn_proj, self.sn_q, self.sn_k, self.sn2_q = (_np(m, device) for m in (a.proj_sn, a.sn_q, a.sn_k, a.sn2_q))
        # q and k read the same spikes: with parameter-free neurons of equal settings (LIF / IF) they run as ONE fused GEMM
        # over the stacked weights [Wq; Wk]; the positional term only exists for the k half (q adds zeros)
        self.qk = None
        sq, sk = self.sn_q, self.sn_k
        if sq.kind != "psn" and (sq.kind, sq.tau, sq.v_th, sq.v_reset) == (sk.kind, sk.tau, sk.v_th, sk.v_reset):
            w = torch.cat([a.linear_q.weight.detach().float(), a.linear_k.weight.detach().float()], 0).to(device).contiguous()
            self.qk = {"Wp": hip.split_weight(w, nsplit), "digits": hip.split_weight_i8x3(w) if self.q.digits is not None else None,
                       "alpha": torch.cat([self.q.alpha, self.k.alpha]).contiguous(),
                       "beta": torch.cat([self.q.beta, self.k.beta]).contiguous(),
                       # the positional table is consumed as flat (Tq, N1, C) memory (the reference's raw reshape, :678-679)
                       "add": torch.cat([torch.zeros((self.pe.numel() // self.q.N, self.q.N), device=device),
                                         self.pe.reshape(-1, self.q.N)], -1).contiguous()}
        m = blk.mlp
        self.fc1 = _Lin(m.fc1, m.bn1.norm_layer, device, nsplit)
        self.fc2 = _Lin(m.fc2, m.bn2.norm_layer, device, nsplit)
        if self.fc2.digits is not None and self.fc2.K >= 1536:      # wide stages: fc2 (K = 4 C against few tokens) runs on the small-M kernel
            self.fc2.digits_tiled = hip.tile_weight_i8x3(self.fc2.digits)
        self.sn1, self.sn2 = _np(m.sn1, device), _np(m.sn2, device)


def _conv_planes(w, nsplit, cin_pad=None):
    return hip.pack_conv_weight(w.detach().float(), nsplit, cin_pad)


def _conv_digits(w, nsplit):
    """int8 digit planes of a 3x3 convolution: on 96 input channels what the weight-resident kernel reads for large launches
    (csrc/spike_conv_wres.hip), on multiples of 64 what the small-M kernel reads (csrc/ms_smallm.hip: the U-Net bottleneck); None where
    neither has an instantiation or the exact 3-plane / 1-plane modes were asked for."""
    if nsplit != 2 or tuple(w.shape[2:]) != (3, 3) or w.shape[0] % 32 or not (w.shape[1] in (48, 96) or w.shape[1] % 64 == 0):
        return None
    return hip.pack_conv_weight_i8x3(w.detach().float(), tiled=w.shape[1] not in (48, 96))   # (fragment order: only the small-M kernel reads those)


class _ResBlock:
    """MS_ResBlock weights for the spike-convolution path (3x3, pad 1, NHWC)."""

    def __init__(self, rb, device, nsplit, name=""):
        self.name = name
        self.C = rb.conv1[0].weight.shape[0]
        self.w1, self.w2 = _conv_planes(rb.conv1[0].weight, nsplit), _conv_planes(rb.conv2[0].weight, nsplit)
        self.w1.digits, self.w2.digits = _conv_digits(rb.conv1[0].weight, nsplit), _conv_digits(rb.conv2[0].weight, nsplit)
        self.bn1, self.bn2 = bn_affine(rb.norm1.norm_layer, device), bn_affine(rb.norm2.norm_layer, device)
        self.sn1, self.sn2 = _np(rb.sn1, device), _np(rb.sn2, device)


_DECONV_ROWMAPS = {}


def deconv_classes(w, imgs, H, W, cin_pad, nsplit, device):
    """ConvTranspose2d(k=3, s=2, p=1, output_padding=1) as four output-parity classes.

    out[2y'+py, 2x'+px] only receives the taps whose kernel index has the right parity
    (oy = 2*iy - 1 + ky): py = 0 -> (ky=1, iy=y'); py = 1 -> (ky=2, iy=y') and (ky=0, iy=y'+1); same in x.
    Each class is a 1- or 2-tap implicit GEMM over the INPUT grid whose rows are scattered to the class's
    output pixels by an int32 row map.  w is the reference layout (Cin, Cout, 3, 3)."""
    Cin, Cout = w.shape[:2]
    taps = {0: [(0, 1)], 1: [(0, 2), (1, 0)]}            # parity -> [(input offset, kernel index)]
    out = []
    for py in (0, 1):
        for px in (0, 1):
            ty, tx = taps[py], taps[px]
            wk = torch.zeros((Cout, len(ty), len(tx), cin_pad), dtype=torch.float32, device=w.device)
            for a, (_, ky) in enumerate(ty):
                for b, (_, kx) in enumerate(tx):
                    wk[:, a, b, :Cin] = w.detach().float()[:, :, ky, kx].t()
            key = (imgs, H, W, py, px, str(device))
            if key not in _DECONV_ROWMAPS:
                i = torch.arange(imgs).view(-1, 1, 1)
                y = torch.arange(H).view(1, -1, 1)
                x = torch.arange(W).view(1, 1, -1)
                rm = ((i * 2 * H + 2 * y + py) * 2 * W + 2 * x + px).reshape(-1).to(torch.int32)
                _DECONV_ROWMAPS[key] = rm.to(device)
            out.append({"Wp": hip.split_weight(wk.reshape(Cout, -1), nsplit), "KH": len(ty), "KW": len(tx),
                        "dy": [t[0] for t in ty], "dx": [t[0] for t in tx], "rowmap": _DECONV_ROWMAPS[key]})
    return out


def _pad16(c):
    return (c + 15) // 16 * 16


def _pad32(c):
    return (c + 31) // 32 * 32


def deconv_tap_weights(w, cin_pad, nsplit):
    """ConvTranspose2d weight (Cin, Cout, 3, 3) -> planes of the stacked per-tap matrices (9*Cout, cin_pad): row
    (ky*3+kx)*Cout + co holds w[:, co, ky, kx] (zero for padded input channels)."""
    Cin, Cout = w.shape[:2]
    wk = torch.zeros((9 * Cout, cin_pad), dtype=torch.float32, device=w.device)
    wk[:, :Cin] = w.detach().float().permute(2, 3, 1, 0).reshape(9 * Cout, Cin)
    planes = hip.split_weight(wk, nsplit)
    if nsplit == 2 and cin_pad % 16 == 0:
        dg = hip.split_weight_i8x3(wk)
        if cin_pad % 64 == 0 and cin_pad >= 1024:            # (the small-M kernel's operand: the same matrix as int8 digits in fragment order)
            planes.digits = hip.tile_weight_i8x3(dg)
        elif cin_pad <= 1024:                                # (row-major digits: the weight-resident row-loop kernel, csrc/ms_res.hip)
            planes.digits_rm = dg
    return planes


class MSFlowEngine:
    @classmethod
    def bare(cls, device, nsplit=2):
        """An engine without a model: the helpers (index maps, kernels, tape) for module-level `forward()` calls - the
        block-level classes of the module tree pack themselves and run through the same methods the model does."""
        e = cls.__new__(cls)
        if torch.device(device).type != "cuda":
            raise hip.SdfError("module forwards run on the MI355X HIP engine only (no CPU fallback): move the module and its input to 'cuda'")
        hip.lib()
        e.device, e.nsplit, e._maps, e._deconv, e.tape, e._masks, e.scores = torch.device(device), nsplit, {}, {}, None, {}, None
        e.replicas = False
        return e

    def __init__(self, model):
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise hip.SdfError("the SDformerFlow forward runs on the MI355X HIP engine only: call model.to('cuda') "
                               "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")
        hip.lib()
        dev, ns = self.device, model.gemm_nsplit
        self.nsplit = ns
        unet = model.sttmultires_unet
        sw = unet.encoders.swin3d
        pe = sw.patch_embed
        self.num_bins, self.num_steps = pe.num_bins, pe.num_steps
        self.head_w = pe.head.conv[0].weight.detach().contiguous(memory_format=torch.channels_last)
        self.head_w_oihw = pe.head.conv[0].weight.detach().float().contiguous()
        self.head_bn, self.head_sn = bn_affine(pe.head.norm_layer.norm_layer, dev), _np(pe.head.sn, dev)
        self.conv_w, self.conv_bn = _conv_planes(pe.conv.conv[0].weight, ns), bn_affine(pe.conv.norm_layer.norm_layer, dev)
        self.conv_w.digits = _conv_digits(pe.conv.conv[0].weight, ns)      # 48 -> 96 at stride 2: the digit kernel's even / odd halo form
        U = "sttmultires_unet."
        self.pe_name = U + "encoders.swin3d.patch_embed."
        self.pe_res = [_ResBlock(rb, dev, ns, self.pe_name + f"residual_encoding.resblocks.{i}.") for i, rb in enumerate(pe.residual_encoding.resblocks)]
        self.proj_res_w = pe.proj.conv_res.weight.detach().contiguous(memory_format=torch.channels_last)
        self.proj_res_w2 = pe.proj.conv_res.weight.detach().float().reshape(pe.proj.conv_res.weight.shape[0], -1).contiguous()
        self.proj_res_b = None if pe.proj.conv_res.bias is None else pe.proj.conv_res.bias.detach().float().contiguous()
        self.proj_w = _conv_planes(pe.proj.conv.weight, ns)
        self.proj_w.digits = _conv_digits(pe.proj.conv.weight, ns)         # 96 -> C at stride 2: the digit kernel's two-channel-pass form
        self.proj_bn, self.proj_sn = bn_affine(pe.proj.norm_layer, dev), _np(pe.proj.sn, dev)
        self._maps, self._deconv = {}, {}
        self.tape = None            # parity tests set a list: every neuron layer's spikes are recorded (see _rec)
        self.scores = None          # `log=True`: a list that receives the attention score of the last block of every stage
        self.replicas = False       # forward(..., replicas=True): the batch is B independent batch-1 forwards (see forward)
        self._init_stages(model, unet, sw, dev, ns, U)

    def _init_stages(self, model, unet, sw, dev, ns, U):
        """Everything behind the patch embedding (the SEW engine packs its own: engine_sew.SEWFlowEngine)."""
        self.stages, self.merges = [], []
        for li, layer in enumerate(sw.layers):
            self.stages.append([_Block(b, dev, ns, U + f"encoders.swin3d.layers.{li}.swin_blocks.{bi}.") for bi, b in enumerate(layer.swin_blocks)])
            if layer.downsample is not None:
                d = layer.downsample
                self.merges.append((_Lin(d.reduction, d.norm.norm_layer, dev, ns), _np(d.sn, dev)))
        self.unet_res = [_ResBlock(rb, dev, ns, U + f"resblocks.{i}.") for i, rb in enumerate(unet.resblocks)]
        self.decoders = [(d.deconv[0].weight.detach(), bn_affine(d.norm_layer.norm_layer, dev), _np(d.sn, dev)) for d in unet.decoders]
        # flow prediction = 1x1 convolution with 2 output channels on spikes: the spike GEMM with the weight rows padded
        # to one 32-column block (columns 2.. are zero and never read back)
        self.preds = []
        for p in unet.preds:
            w2 = p.conv[0].weight.detach().float().reshape(p.conv[0].weight.shape[0], -1)        # (2, Cin)
            wp = torch.zeros((32, w2.shape[1]), dtype=torch.float32, device=dev)
            wp[:w2.shape[0]] = w2
            bp = torch.zeros(32, dtype=torch.float32, device=dev)
            bp[:w2.shape[0]] = p.conv[0].bias.detach().float()
            self.preds.append((hip.split_weight(wp, ns), bp, _np(p.sn, dev), w2.shape[0], w2.to(dev).contiguous(),
                               p.conv[0].bias.detach().float().to(dev).contiguous()))

    # ------------------------------------------------------------------ helpers
    def _slice_map(self, B, D, H, W, ws, ss):
        key = ("win", B, D, H, W, ws, ss, self.replicas)
        if key not in self._maps:
            if self.replicas and B > 1:
                # B independent batch-1 problems in one launch sequence: every sample keeps ITS batch-1 window view (hip.replica_slice_map)
                m1, nW = self._slice_map_1(D, H, W, ws, ss)
                if nW % ws[0]:
                    # (the head scramble of a sample must stay on one side of its attention-step boundary: hip.replica_zsrc_map, zg_rep)
                    raise hip.ReplicaGeometryError(f"replicas need a window count per sample ({nW}) that is a multiple of the window depth ({ws[0]})")
                self._maps[key] = (hip.replica_slice_map(m1, nW, B, ws[0], ws[1] * ws[2], D * H * W), B * nW)
            else:
                self._maps[key] = hip.window_slice_map(B, D, H, W, ws, ss, self.device)    # built on the device, cached per shape
        return self._maps[key]

    def _slice_map_1(self, D, H, W, ws, ss):
        key = ("win", 1, D, H, W, ws, ss, False)
        if key not in self._maps:
            self._maps[key] = hip.window_slice_map(1, D, H, W, ws, ss, self.device)
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

    @staticmethod
    def _check_cl(x):
        """The kernels address activations as dense channel-last (B,D,h,w,C) fp32 buffers (and update them in place)."""
        if x.dim() != 5 or not x.is_contiguous() or x.dtype != torch.float32:
            raise hip.SdfError(f"expected a contiguous fp32 (B,D,h,w,C) tensor, got shape {tuple(x.shape)} strides {x.stride()} "
                               f"{x.dtype}; call .contiguous() (the update is in place)")

    def _rec(self, name, spikes, layout):
        """Parity tape (tests only; `self.tape = []` switches it on, never during graph capture): a copy of the u8 spikes of
        neuron layer `name` (the state_dict prefix of its Spiking_neuron); `layout` says how the copy maps onto the reference's
        tensor at that call: "flat" = same memory order (reshape), "BDHWC->TBCHW" / "BDHWC->TBHWC" = permute."""
        if self.tape is not None:
            self.tape.append((name, spikes.clone(), layout))

    def _neuron_bd(self, x, p, bn=None, out_dtype=torch.uint8):
        """Neuron over D of a channel-last (B,D,h,w,C) activation, BN fused when given."""
        self._check_cl(x)
        B, D, h, w, Cc = x.shape
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
        a, b = bn if bn is not None else (None, None)
        n = h * w * Cc
        hip.neuron_fwd(x, out, D, B, n, D * n, n, D * n, n, p, alpha=a, beta=b, Cch=Cc, inner=1)
        return out

    def _conv3x3(self, s, Wp, Cout, stride=1, bn=None, resid=None, sn=None, membrane=False, _dst=None):
        """3x3 / pad 1 spike convolution on (B,D,h,w,Cin) u8 -> (B,D,oh,ow,Cout): fp32 (BN, + resid) or, with `sn`,
        spikes of the fused BN + neuron over D; with `membrane` as well, (fp32 BN + resid, spikes of SN(that)) from one launch."""
        B, D, h, w, Cin = s.shape
        oh, ow = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        a, b = bn if bn is not None else (None, None)
        # large 3x3 / stride-1 launches on 96 channels: int8 digit planes, weights resident in LDS (csrc/spike_conv_wres.hip)
        digits = getattr(Wp, "digits", None)
        if self.replicas and B > 1 and _dst is None and digits is None:
            # 16-bit planes only (no digit form of this layer): the streaming kernels' tile family and split-K plan follow the row count
            return self._conv3x3_chunks(s, Wp, Cout, stride, bn, resid, sn, membrane, 1)
        if self.replicas and B > 1 and _dst is None and digits is not None:
            # replicas: every layer must take the kernel family (and weight representation) its batch-1 forward takes, whatever the
            # size rules say about B-fold rows - else the flows are not the batch-1 flows bit for bit.  Where the rules answer
            # differently for B samples, the largest sample chunk that is routed like ONE sample runs per launch.
            route = lambda n: (hip.smallm_conv_applicable(n * D, h, w, Cin, Cout, stride, D), hip.conv_wres_applicable(n * D, h, w, Cin, Cout, stride, 1),
                               hip.conv_wres_applicable(n * D, h, w, Cin, Cout, stride, D))
            r1 = route(1)
            if route(B) != r1:
                bc = max(n for n in range(1, B) if B % n == 0 and route(n) == r1)
                return self._conv3x3_chunks(s, Wp, Cout, stride, bn, resid, sn, membrane, bc)
        if digits is not None and (sn is None or sn.kind in ("lif", "if", "psn")) and hip.smallm_conv_applicable(B * D, h, w, Cin, Cout, stride, D):
            # few rows against many weights (the U-Net bottleneck: 1 080 rows x 768 x 6 912): one launch, K split over the waves of a
            # workgroup, sum + BN + shortcut + neuron in its epilogue (csrc/ms_smallm.hip)
            out = None if sn is not None and not membrane else torch.empty((B, D, oh, ow, Cout), dtype=torch.float32, device=s.device)
            sp = torch.empty((B, D, oh, ow, Cout), dtype=torch.uint8, device=s.device) if sn is not None else None
            try:
                hip.spike_conv2d(s, digits, B * D, h, w, Cin, oh, ow, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=out, out_spike=sp, alpha=a, beta=b,
                                 resid=resid if (membrane or sn is None) else None, sn=sn, sn_T=D if sn is not None else 0,
                                 pos=(B * oh * ow, oh * ow, D * oh * ow, oh * ow))
                return sp if sn is not None and not membrane else ((out, sp) if sn is not None else out)
            except hip.SdfError as e:
                # the Python mirror `smallm_conv_applicable` admits a shape the library's own rule refuses (size / alignment, 31-bit sizes,
                # an unusual tau: ADVICE r4 / r5): the 16-bit planes are still here - the general path below serves it
                if e.rc not in (hip.E_SHAPE, hip.E_ALIGN):
                    raise
                if sn is not None and sn.kind == "psn":
                    # a PSN res-block was sent here because the small-M kernel fuses its neuron; refused, it runs as the fp32
                    # convolution (streaming kernels, 16-bit planes) + the neuron kernel, not on the streaming kernels' fused epilogue
                    m = torch.empty((B, D, oh, ow, Cout), dtype=torch.float32, device=s.device)
                    hip.spike_conv2d(s, Wp, B * D, h, w, Cin, oh, ow, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=m, alpha=a, beta=b,
                                     resid=resid if membrane else None)
                    sp = self._neuron_bd(m, sn)
                    return (m, sp) if membrane else sp
        if digits is not None and B > 1 and not hip.conv_wres_applicable(B * D, h, w, Cin, Cout, stride, 1):
            # too large for the digit kernel's 31-bit operand offsets as one launch (configs[4]: 80 images of 240 x 320 x 96 fp32):
            # batch elements are independent - the largest batch chunk that fits runs per launch, writing into its slice of the outputs
            bc = self._digit_chunk(B, D, h, w, Cin, Cout, stride)
            if bc:
                return self._conv3x3_chunks(s, Wp, Cout, stride, bn, resid, sn, membrane, bc)
        if digits is not None and sn is not None and (sn.kind == "psn" or D not in (5, 10, 20)) and \
                hip.conv_wres_applicable(B * D, h, w, Cin, Cout, stride, 1):
            # the digit kernel's fused form is LIF / IF over T = 5 / 10 / 20; other neurons (the shipped PSN) take its fp32 form and
            # the neuron kernel behind it - still ahead of the streaming kernel's fused epilogue (profiles/r2j_psn.txt)
            m = self._conv3x3(s, Wp, Cout, stride, bn=bn, resid=resid if membrane else None)
            sp = self._neuron_bd(m, sn)
            return (m, sp) if membrane else sp
        if digits is not None and (sn is None or (sn.kind != "psn" and D in (5, 10, 20))) and \
                hip.conv_wres_applicable(B * D, h, w, Cin, Cout, stride, D if sn is not None else 1):
            Wp = digits                                               # (the kernel's fused form: LIF / IF over T = 5 / 10 / 20)
        if sn is None:
            out = _dst[0] if _dst is not None and len(_dst) == 1 and _dst[0].dtype == torch.float32 else \
                torch.empty((B, D, oh, ow, Cout), dtype=torch.float32, device=s.device)
            hip.spike_conv2d(s, Wp, B * D, h, w, Cin, oh, ow, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=out, alpha=a, beta=b,
                             resid=resid)
            return out
        if _dst is not None and len(_dst) == (2 if membrane else 1) and _dst[-1].dtype == torch.uint8:
            out, m = _dst[-1], (_dst[0] if membrane else None)
        else:
            out = torch.empty((B, D, oh, ow, Cout), dtype=torch.uint8, device=s.device)
            m = torch.empty((B, D, oh, ow, Cout), dtype=torch.float32, device=s.device) if membrane else None
        hip.spike_conv2d(s, Wp, B * D, h, w, Cin, oh, ow, 3, 3, stride, (-1, 0, 1), (-1, 0, 1), out=m, out_spike=out, alpha=a, beta=b,
                         resid=resid if membrane else None, sn=sn, sn_T=D, pos=(B * oh * ow, oh * ow, D * oh * ow, oh * ow))
        return (m, out) if membrane else out

    def _conv3x3_chunks(self, s, Wp, Cout, stride, bn, resid, sn, membrane, bc):
        """_conv3x3 in chunks of bc batch elements (batch elements are independent), every chunk writing its slice of the outputs."""
        B, D, h, w, _ = s.shape
        oh, ow = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
        kinds = [torch.float32] if sn is None else ([torch.float32, torch.uint8] if membrane else [torch.uint8])
        outs = tuple(torch.empty((B, D, oh, ow, Cout), dtype=k, device=s.device) for k in kinds)
        if resid is not None:
            resid = resid.view(B, D, oh, ow, Cout)                # (callers also hand it over as (B * D, oh, ow, Cout) images)
        for b0 in range(0, B, bc):
            dst = tuple(o[b0:b0 + bc] for o in outs)
            r = self._conv3x3(s[b0:b0 + bc], Wp, Cout, stride, bn, None if resid is None else resid[b0:b0 + bc], sn, membrane, dst)
            r = r if isinstance(r, tuple) else (r,)
            for o, t in zip(dst, r):
                if t.data_ptr() != o.data_ptr():                  # (the two-launch PSN form allocates its own)
                    o.copy_(t)
        return outs if len(outs) > 1 else outs[0]

    def _rb(self, B):
        """The batch size the ROUTING rules are asked about: with replicas every layer is routed like its batch-1 forward."""
        return 1 if self.replicas else B

    @staticmethod
    def _digit_chunk(B, D, h, w, Cin, Cout, stride):
        """Largest number of batch elements per launch the digit kernel addresses, chosen among the DIVISORS of B so that every chunk is
        the same launch (a ragged last chunk could miss the kernel's "enough tiles" rule and fall to a streaming kernel that has no T = 5 / 20
        form); 0: none."""
        for bc in range(B - 1, 0, -1):
            if B % bc == 0 and hip.conv_wres_applicable(bc * D, h, w, Cin, Cout, stride, 1):
                return bc
        return 0

    @staticmethod
    def _fusable(B, D, h, w, C, Wp=None):
        """The fused-neuron convolution epilogue pays off once the tiles fill the chip.  The streaming kernels have it for T = 10; the
        weight-resident digit kernel (96 -> 96 channels, `Wp.digits`) rolls its time loop and takes T = 5 / 20 as well (configs[4])."""
        big = (B * D * h * w + 255) // 256 * (C // 96) >= 128
        if D == 10:
            return big
        if not (big and D in (5, 20) and C == 96 and getattr(Wp, "digits", None) is not None and Wp.shape[-1] == 9 * 96):
            return False
        bc = B if hip.conv_wres_applicable(B * D, h, w, 96, C, 1, 1) else MSFlowEngine._digit_chunk(B, D, h, w, 96, C, 1)
        return bc > 0 and hip.conv_wres_applicable(bc * D, h, w, 96, C, 1, D)

    def _resblock(self, m, rb, s1=None, next_sn=None):
        """MS_ResBlock on a (B,D,h,w,C) membrane: SN -> conv+BN+SN (one kernel) -> conv+BN+identity (one kernel)
        (reference Spiking_modules.py:906-933).  `s1` = SN1(m) if the producer of m already made it; with `next_sn` the
        second convolution also emits the spikes of the NEXT layer's neuron on its output: returns (m', SN_next(m'))."""
        B, D, h, w, _ = m.shape
        if s1 is None:
            s1 = self._neuron_bd(m, rb.sn1)
        self._rec(rb.name + "sn1.spiking_neuron.", s1, "BDHWC->TBCHW")
        fus = self._fusable(self._rb(B), D, h, w, rb.C, rb.w1) or \
            (getattr(rb.w1, "digits", None) is not None and rb.sn2.kind in ("lif", "if", "psn") and hip.smallm_conv_applicable(self._rb(B) * D, h, w, rb.C, rb.C, 1, D))
        if fus:
            s2 = self._conv3x3(s1, rb.w1, rb.C, bn=rb.bn1, sn=rb.sn2)
        else:       # few rows (U-Net bottleneck): the fp32 epilogue can split K over the chip; neuron as its own launch
            s2 = self._neuron_bd(self._conv3x3(s1, rb.w1, rb.C), rb.sn2, bn=rb.bn1)
        self._rec(rb.name + "sn2.spiking_neuron.", s2, "BDHWC->TBCHW")
        if next_sn is not None and fus:
            return self._conv3x3(s2, rb.w2, rb.C, bn=rb.bn2, resid=m, sn=next_sn, membrane=True)
        m2 = self._conv3x3(s2, rb.w2, rb.C, bn=rb.bn2, resid=m)
        return (m2, self._neuron_bd(m2, next_sn)) if next_sn is not None else m2

    # ------------------------------------------------------------------ stages (each usable stand-alone in tests)
    def patch_embed(self, x):
        """(B,bins,2,H,W) -> membrane (B,D,H/4,W/4,C), channel-last (reference Spiking_modules.py:1770-1790)."""
        xv = x                                                   # the voxel as handed over (the kernel reads its first num_bins bins in place)
        if x.size(1) > self.num_bins:
            x = x[:, :self.num_bins]
        B, T = x.shape[0], self.num_steps
        H, W = x.shape[-2:]
        num_ch = self.num_bins * 2 // T
        # head: real-valued 2-channel input -> conv + BN + SN as one kernel that reads the voxel in place (channel ci of step t is
        # polarity ci % 2 of bin (ci // 2) * T + t; shapes outside its build: re-layout + library convolution, then BN + SN
        # fused in the neuron kernel)
        if hip.head_conv_sn_supported(T, H, W, num_ch, self.head_w_oihw.shape[0]) and xv.is_contiguous():
            s = hip.head_conv_sn(xv, self.head_w_oihw, B, T, H, W, self.head_sn, alpha=self.head_bn[0], beta=self.head_bn[1],
                                 voxel_bins=xv.shape[1])
        else:
            ev = x.permute(0, 2, 3, 4, 1)                                                 # (B,2,H,W,bins)
            xr = torch.stack([ev[:, i % 2, :, :, (i // 2) * T:(i // 2 + 1) * T] for i in range(num_ch)], -1)   # (B,H,W,T,ch)
            xr = xr.permute(0, 3, 1, 2, 4).contiguous().view(B * T, H, W, num_ch)           # NHWC, image = (b,t)
            y = F.conv2d(xr.permute(0, 3, 1, 2), self.head_w, None, 1, 1).contiguous(memory_format=torch.channels_last)
            s = self._neuron_bd(y.permute(0, 2, 3, 1).view(B, T, H, W, -1), self.head_sn, bn=self.head_bn)
        self._rec(self.pe_name + "head.sn.spiking_neuron.", s, "BDHWC->TBCHW")
        # every membrane of the patch embedding leaves its convolution together with the spikes of the neuron that reads it
        sns = [rb.sn1 for rb in self.pe_res] + [self.proj_sn]
        C0 = self.conv_w.shape[1]
        oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        fus = self._fusable(self._rb(B), T, oh, ow, C0)
        if not fus and T in (5, 20) and getattr(self.conv_w, "digits", None) is not None and sns[0].kind != "psn":
            # T = 5 / 20 (configs[4]): the digit kernel's stride-2 form rolls its time loop as the stride-1 form does (_fusable's rule)
            c0 = s.shape[-1]
            Br = self._rb(B)
            bc = Br if hip.conv_wres_applicable(Br * T, H, W, c0, C0, 2, 1) else self._digit_chunk(Br, T, H, W, c0, C0, 2)
            fus = bc > 0 and hip.conv_wres_applicable(bc * T, H, W, c0, C0, 2, T)
        if fus:
            m, s1 = self._conv3x3(s, self.conv_w, C0, stride=2, bn=self.conv_bn, sn=sns[0], membrane=True)
        else:
            m = self._conv3x3(s, self.conv_w, C0, stride=2, bn=self.conv_bn)
            s1 = self._neuron_bd(m, sns[0])
        for i, rb in enumerate(self.pe_res):
            m, s1 = self._resblock(m, rb, s1=s1, next_sn=sns[i + 1])
        # PED projection: 1x1 stride-2 shortcut on the (real-valued) membrane + SN -> conv3x3 s2 -> BN, summed in the epilogue
        Bm, Dm, h, w, Cc = m.shape
        self._rec(self.pe_name + "proj.sn.spiking_neuron.", s1, "BDHWC->TBCHW")
        if hip.pointwise_conv_supported(Cc, self.proj_res_w2.shape[0]):                # exact fp32 MFMA (csrc/pointwise_conv.hip)
            res = hip.pointwise_conv_f32(m.view(B * T, h, w, Cc), self.proj_res_w2, 2, self.proj_res_b)
        else:
            res = F.conv2d(m.view(B * T, h, w, Cc).permute(0, 3, 1, 2), self.proj_res_w, self.proj_res_b, 2)
            res = res.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        return self._conv3x3(s1, self.proj_w, self.proj_w.shape[1], stride=2, bn=self.proj_bn, resid=res)

    def _zsrc_map(self, rowmap, B_, Tq, N1, nH, x_rows, key):
        key = ("zsrc", nH, self.replicas) + key
        if key not in self._maps:
            B = key[3]
            if self.replicas and B > 1:
                _, D, H, W, ws, ss = key[3:]
                m1, nW = self._slice_map_1(D, H, W, ws, ss)
                z1 = hip.window_zsrc_map(m1, nW, Tq, N1, nH, x_rows // B)
                self._maps[key] = hip.replica_zsrc_map(z1, nW, B, Tq, N1, nH * 32)
            else:
                self._maps[key] = hip.window_zsrc_map(rowmap, B_, Tq, N1, nH, x_rows)     # built on the device, cached per shape
        return self._maps[key]

    def attention(self, x, blk: _Block, score_out=None, emit=None):
        """x (B,D,H,W,C) += SSA(x), in place (reference Spiking_swin_transformer3D.py:781-821, 661-717, :840).  `score_out` (a list)
        also receives the block's attention score (T', B_, Wh, Ww, C) - `return_attention=True` (:807-808) without abandoning x.
        `emit` = (u8 buffer, neuron): on the wide stages the projection's launch also leaves SN(x) over D there (the MLP's first
        neuron, csrc/ms_wide.hip); `self._emitted` says whether it did."""
        self._check_cl(x)
        B, D, H, W, Cc = x.shape
        ws, ss = get_window_size((D, H, W), blk.window_size, blk.shift_size)
        if tuple(ws) != tuple(blk.window_size):
            # the reference clamps the window (:786) and then cannot view its positional encoding, sized for the nominal window, as
            # (T', 1, Wh, Ww, C) (:678, RuntimeError); refused here as well instead of reading the table with the wrong pitch
            raise hip.SdfError(f"feature map {(D, H, W)} is smaller than the window {tuple(blk.window_size)}: the positional encoding of "
                               "Spiking_QK_WindowAttention3D is defined for the nominal window only (reference Spiking_swin_transformer3D.py:678)")
        rowmap, B_ = self._slice_map(B, D, H, W, ws, ss)
        Tq, N1 = ws[0], ws[1] * ws[2]
        rep_windows = B_ // B if self.replicas and B > 1 else 0
        if rep_windows and (self.tape is not None or score_out is not None):
            raise hip.SdfError("the parity tape and log=True follow one forward: run them without replicas")
        # one C-ABI call: neuron over the gathered slices -> q|k spike GEMM (+BN, +PE, neurons fused) -> token gate ->
        # projection spike GEMM through the head scramble with bias + BN + scatter + residual (csrc/qk_attn.hip)
        keep = [] if score_out is not None else None
        if self.tape is not None:
            # the slice spikes are overwritten by the gate inside the call: the tape runs the same kernel on the same input first
            rows, keep = B_ * N1, []
            xs = torch.empty((Tq, rows, Cc), dtype=torch.uint8, device=x.device)
            hip.neuron_fwd(x, xs, Tq, 1, rows * Cc, 0, 0, 0, rows * Cc, blk.sn_proj, rowmap=rowmap, rowlen=Cc)
            self._rec(blk.name + "attn.proj_sn.spiking_neuron.", xs, "flat")
        info = {}
        zsrc = self._zsrc_map(rowmap, B_, Tq, N1, blk.nH, B * D * H * W, (B, D, H, W, tuple(ws), tuple(ss))) if Cc >= _WIDE_MINC and Tq == 2 else None
        kw = dict(qk=blk.qk) if blk.qk is not None else dict(q_lin=blk.q, k_lin=blk.k, pe=blk.pe)
        hip.qk_attn(x, rowmap, B_, Tq, N1, blk.nH, blk.p, blk.sn_proj, blk.sn_q, blk.sn_k, blk.sn2_q, keep_ws=keep, x_src=zsrc, emit=emit,
                    info=info, rep_windows=rep_windows, **kw)
        if score_out is not None:
            # the gate's output replaced the slice spikes at the head of the workspace (csrc/qk_attn.hip)
            e = keep[0][:Tq * B_ * N1 * Cc].view(Tq, B_ * N1, Cc)
            score_out.append(attention_score(e, blk.attn_sn, blk.nH, Tq, B_, N1).view(Tq, B_, ws[1], ws[2], Cc))
        if keep and self.tape is not None:
            M = Tq * B_ * N1
            qk = keep[0][(M * Cc + 255) // 256 * 256:][:M * 2 * Cc]
            q, k = (qk.view(M, 2 * Cc)[:, :Cc], qk.view(M, 2 * Cc)[:, Cc:]) if blk.qk is not None else (qk[:M * Cc], qk[M * Cc:])
            self._rec(blk.name + "attn.sn_q.spiking_neuron.", q.reshape(Tq, B_ * N1, Cc), "flat")
            self._rec(blk.name + "attn.sn_k.spiking_neuron.", k.reshape(Tq, B_ * N1, Cc), "flat")
        self._emitted = bool(info.get("emitted"))
        return x

    def mlp(self, x, blk: _Block, ws=None, s1_ready=False, emit_next=None):
        """x (B,D,H,W,C) += MLP(x) over the true time axis D, in place (reference :164-181, :845).  `ws` / `s1_ready`: the caller's
        workspace with SN1's spikes already at its head (wide stages: left there by the attention's projection); `emit_next`: see
        _next_spikes."""
        self._check_cl(x)
        keep = [] if self.tape is not None else None
        hip.ms_mlp(x, blk.fc1, blk.fc2, blk.sn1, blk.sn2, keep_ws=keep, ws=ws, s1_ready=s1_ready, emit_next=emit_next)      # one C-ABI call (csrc/qk_attn.hip: sdf_ms_mlp_fwd)
        if keep:
            B, D, H, W, Cc = x.shape
            tok = B * D * H * W
            self._rec(blk.name + "mlp.sn1.spiking_neuron.", keep[0][:tok * Cc].view(B, D, H, W, Cc), "BDHWC->TBHWC")
            self._rec(blk.name + "mlp.sn2.spiking_neuron.", keep[0][(tok * Cc + 255) // 256 * 256:][:tok * blk.fc1.N].view(B, D, H, W, -1), "BDHWC->TBHWC")
        return x

    def swin_block(self, x, s, i, emit_next=None):
        blk = self.stages[s][i]
        last = self.scores is not None and i == len(self.stages[s]) - 1          # log=True: the last block of every stage (:1090-1105)
        # wide stages: the projection emits the MLP's first spikes (tiled hand-over layout; the tape / score paths keep row-major spikes
        # and let the MLP run its own first neuron)
        ws = hip.ms_mlp_workspace(x, blk.fc1.N) if x.shape[-1] >= _WIDE_MINC and self.tape is None and not last else None
        self.attention(x, blk, self.scores if last else None, emit=(ws, blk.sn1) if ws is not None else None)
        return self.mlp(x, blk, ws=ws, s1_ready=self._emitted, emit_next=emit_next)

    def _tail_kwargs(self):
        """encoder() arguments of forward(): the neuron that reads the last stage's output first (MS_ResBlock.sn1 of the bottleneck)."""
        return {"tail_sn": self.unet_res[0].sn1} if self.unet_res else {}

    def _next_spikes(self, x, blk, sn):
        """(u8 buffer, neuron) for the MLP of `blk` to fill with SN(x after its update) - the first neuron of the layer behind the
        stage - or None where that MLP does not run on the wide-stage kernels (the layer then runs its own neuron)."""
        if x.shape[-1] < _WIDE_MINC or not hip.ms_mlp_is_wide(x, blk.fc1, blk.fc2, blk.sn1, blk.sn2, emit_sn=sn):
            return None
        return torch.empty(x.shape, dtype=torch.uint8, device=x.device), sn

    def patch_merge(self, x, s, packed=None, spikes=None):
        """(B,D,H,W,C) -> (B,D,H/2,W/2,2C) (reference :952-974).  `spikes` = SN(x) (B,D,H,W,C) u8 when the stage's last MLP emitted
        them: the 2x2 concatenation is then index arithmetic inside the reduction GEMM (hip.ms_patch_merge)."""
        lin, sn = self.merges[s] if packed is None else packed
        self._check_cl(x)
        B, D, H, W, Cc = x.shape
        name = f"sttmultires_unet.encoders.swin3d.layers.{s}.downsample.sn.spiking_neuron."
        if spikes is None and getattr(lin, "digits", None) is not None and D in (10, 20) and sn.kind in hip.KIND:
            # the stage's last MLP did not emit them (stage 0: it runs on the one-launch fp16-plane kernel): the neuron as a plain
            # launch over x, then the same digit-plane reduction with the 2x2 concatenation as operand addressing - 8 + 19 us at the
            # first merge of the shipped model against 33 us for the gathered neuron + spike GEMM below
            spikes = self._neuron_bd(x, sn)
        if spikes is not None:
            out = hip.ms_patch_merge(spikes, lin)
            if out is not None:
                if self.tape is not None:                   # the tape holds the reference's (T, B, H/2, W/2, 4C) concatenation
                    src, H2, W2 = merge_row_map(B, D, H, W)
                    idx = torch.from_numpy(src.reshape(-1).astype("int64")).to(x.device)
                    flat = torch.cat([spikes.view(-1, Cc), spikes.new_zeros(1, Cc)], 0)
                    self._rec(name, flat[idx].view(D, B, H2, W2, 4 * Cc), "flat")
                return out
        rowmap, H2, W2, out_map = self._merge_map(B, D, H, W)
        rows = B * H2 * W2
        sp = torch.empty((D * rows, 4 * Cc), dtype=torch.uint8, device=x.device)
        hip.neuron_fwd(x, sp, D, 1, rows * 4 * Cc, 0, 0, 0, rows * 4 * Cc, sn, rowmap=rowmap, rowlen=Cc)
        self._rec(name, sp.view(D, B, H2, W2, 4 * Cc), "flat")
        out = torch.empty((B, D, H2, W2, lin.N), dtype=torch.float32, device=x.device)
        hip.spike_gemm(sp, lin.Wp, out, D * rows, lin.N, 4 * Cc, alpha=lin.alpha, beta=lin.beta, out_rowmap=out_map)
        return out

    def encoder(self, x, tail_sn=None):
        """-> per-stage features, channel-last (B,D,h,w,C) (reference :1223-1246 + Spiking_STSwinNet.py:77-85).  `tail_sn` = the first
        neuron of whatever reads the last stage's output: where the last MLP can emit its spikes they are left in `self.tail_spikes`."""
        y = self.patch_embed(x)
        feats = []
        self.tail_spikes = None
        for s, blocks in enumerate(self.stages):
            nxt_sn = self.merges[s][1] if s < len(self.merges) else tail_sn
            emit = None
            for i in range(len(blocks)):
                if i == len(blocks) - 1 and nxt_sn is not None:
                    emit = self._next_spikes(y, blocks[i], nxt_sn)
                y = self.swin_block(y, s, i, **({"emit_next": emit} if emit is not None else {}))
            feats.append(y)                                 # the merge writes a new tensor; nothing touches y after this
            if s < len(self.merges):
                y = self.patch_merge(y, s, **({"spikes": emit[0]} if emit is not None else {}))
            elif emit is not None:
                self.tail_spikes = emit[0]
        return feats

    def _deconv_classes(self, i, B, D, h, w, cin_pad, wkey="ref", wuse=None):
        key = (i, B, D, h, w, wkey)
        if key not in self._deconv:
            self._deconv[key] = deconv_classes(self.decoders[i][0] if wuse is None else wuse, B * D, h, w, cin_pad, self.nsplit, self.device)
        return self._deconv[key]

    def _decoder_weight(self, i, wkey, C1, C2):
        """ConvTranspose2d weight (Cin, Cout, 3, 3) of decoder i in the reference's input-channel order [pred | y | skip] ("ref")
        or in the physical order of the concatenation-free path [y | skip | pred, 0, 0] ("perm")."""
        wdec = self.decoders[i][0]
        if wkey == "ref" or i == 0:
            return wdec
        key = ("wperm", i)
        if key not in self._deconv:
            n = wdec.shape[0] - C1 - C2                                   # prediction channels in front (2)
            pad = torch.zeros((4 - n,) + tuple(wdec.shape[1:]), dtype=wdec.dtype, device=wdec.device)
            self._deconv[key] = torch.cat([wdec[n:n + C1], wdec[n + C1:], wdec[:n], pad], 0).contiguous()
        return self._deconv[key]

    def _decoder_geometry(self, i, B, D, h, w, cin):
        """(as_gemm, padded channel count of the NHWC spike image) of decoder level i on an (h, w) input with cin channels.
        Small levels: ONE plain spike GEMM over the nine stacked tap matrices + a col2im pass fills the chip; the four parity-class
        convolutions (no 9x intermediate) are kept where that intermediate would cost more than it saves."""
        cout = self.decoders[i][0].shape[1]
        as_gemm = self._rb(B) * D * h * w * 9 * cout * 4 <= 64 << 20
        return as_gemm, (_pad32(cin) if as_gemm else _pad16(cin))

    def _prepare_decoder_images(self, feats, y0, out_size):
        """NHWC u8 spike images of every decoder level with the skip slices written by one multi-descriptor neuron launch, or None
        when the pyramid is not regular / a level's head has no one-launch kernel (the per-level path handles those)."""
        E = len(feats)
        if out_size is None or E < 2 or E > 6:
            return None
        B, D, h0, w0, _ = feats[-1].shape
        if tuple(y0.shape[2:4]) != (h0, w0):
            return None
        imgs, calls, c1 = [], [], y0.shape[-1]
        for i in range(E):
            skip = feats[E - 1 - i]
            h, w = h0 << i, w0 << i
            if tuple(skip.shape[2:4]) != (h, w):
                return None
            wdec, bn, sn = self.decoders[i]
            cout = wdec.shape[1]
            pw, pb, psn, nout, w2, b2 = self.preds[i]
            ok = nout == 2 and out_size[0] % (2 * h) == 0 and out_size[1] % (2 * w) == 0 and \
                hip.pred_head_supported(D, cout, out_size[0], out_size[1], 2 * h, 2 * w, psn, self.decoders[i + 1][2] if i + 1 < E else None)
            if not ok:
                return None
            C2 = skip.shape[-1]
            cin = c1 + C2 + (4 if i > 0 else 0)
            _, cp = self._decoder_geometry(i, B, D, h, w, cin)
            img = torch.empty((B, D, h, w, cp), dtype=torch.uint8, device=y0.device)
            if i == 0 and cp != cin:
                img[..., cin:].zero_()                                  # (levels >= 1: zeroed by the head of the level above)
            hw = h * w
            # (the batch elements are the descriptor's outermost dimension: one multi-descriptor launch for the whole batch)
            calls.append((skip, img.view(-1)[c1:], D, hw, C2, C2, hw * C2, cp, hw * cp, sn, None, 0, None, None, 0, 1, None, 0, 0, None,
                          (B, D * hw * C2, D * hw * cp)))
            imgs.append(img)
            c1 = cout
        # level 0's own input y rides in the same launch (its neuron is the level's `sn` too; six descriptors at most)
        self._ready_has_y = len(calls) < 6
        if self._ready_has_y:
            cy, hw0, cp0 = y0.shape[-1], h0 * w0, imgs[0].shape[-1]
            calls.append((y0, imgs[0].view(-1), D, hw0, cy, cy, hw0 * cy, cp0, hw0 * cp0, self.decoders[0][2], None, 0, None, None, 0, 1, None, 0, 0,
                          None, (B, D * hw0 * cy, D * hw0 * cp0)))
        hip.neuron_multi_fwd(calls)
        return imgs

    def unet_tail(self, feats, out_size=None, s1=None):
        """res-blocks + decoders + per-scale predictions on channel-last (B,D,h,w,C) features
        (reference Spiking_STSwinNet.py:161-182).  `s1` = the first res-block's SN1(feats[-1]) where the encoder's last MLP emitted it
        (encoder(..., tail_sn=)).  Returns the per-scale predictions (B,D,h',w',2) fp32; with `out_size` = (H, W)
        the flow maps (time sum + nearest upsampling, reference :289-303) of the levels whose prediction head ran as one launch
        are left in `self._flows` (None where the three-launch form ran: forward() calls sdf_flow_out_fwd for those)."""
        y = feats[-1]
        for i, rb in enumerate(self.unet_res):                # a block's second convolution also emits the next block's first spikes
            nxt = self.unet_res[i + 1].sn1 if i + 1 < len(self.unet_res) else None
            r = self._resblock(y, rb, s1=s1, next_sn=nxt)
            y, s1 = r if nxt is not None else (r, None)
        preds, E = [], len(feats)
        self._flows = [None] * E
        carried = None          # this level's spike image with the [y | prediction] slices already written by the previous level's head
        # Regular pyramid (every level twice the size of the one before, every prediction head on its one-launch kernel): the
        # spike images of ALL levels are laid out now and the decoders' neuron on the four skip tensors - four small launches
        # before - is ONE launch (sdf_neuron_multi_fwd); the [y | prediction] slices arrive from the level above's head.
        ready = self._prepare_decoder_images(feats, y, out_size) if self.tape is None else None
        for i in range(E):
            skip = feats[E - 1 - i]
            B, D, h, w, _ = skip.shape
            # skip_concat(pred, skip_concat(y, skip)) on channels (:168-172); the prediction has y's size (both come from z)
            wdec, bn, sn = self.decoders[i]
            cout = wdec.shape[1]
            same = tuple(y.shape[2:4]) == (h, w)
            if same:
                # no concatenation: the decoder's neuron runs on each source and writes its spikes straight into that source's
                # channel slice of the NHWC spike image (physical channel order [y | skip | prediction padded to 4]; the
                # transposed-convolution weight rows are permuted to match once, at pack time)
                C1, C2 = y.shape[-1], skip.shape[-1]
                cin = C1 + C2 + (4 if i > 0 else 0)
                as_gemm, cp = self._decoder_geometry(i, B, D, h, w, cin)
                if ready is not None:
                    s, c0 = ready[i], 0                                   # skip slice (and padding) already in place; level 0 takes y now
                    srcs = [(y, C1, C1)] if i == 0 and not self._ready_has_y else []
                elif carried is not None:
                    s, srcs, c0 = carried, [(skip, C2, C2)], C1          # [y | . | prediction | zeros] came from the level above
                else:
                    s = (torch.zeros if cp != cin else torch.empty)((B, D, h, w, cp), dtype=torch.uint8, device=y.device)
                    srcs, c0 = [(y, C1, C1), (skip, C2, C2)] + ([(preds[-1], 4, preds[-1].shape[-1])] if i > 0 else []), 0
                hw = h * w
                for src, take, pitch in srcs:                             # (tensor, channels taken, pitch)
                    for b in range(B):
                        hip.neuron_fwd(src[b], s[b].view(-1)[c0:], D, hw, take, pitch, hw * pitch, cp, hw * cp, sn)
                    c0 += take
                wkey = "perm"
                if self.tape is not None:             # the reference's channel order is [prediction | y | skip]
                    npred = self.preds[i - 1][3] if i > 0 else 0
                    self._rec(f"sttmultires_unet.decoders.{i}.sn.spiking_neuron.",
                              torch.cat([s[..., C1 + C2:C1 + C2 + npred], s[..., :C1 + C2]], -1), "BDHWC->TBCHW")
            else:
                parts = ([preds[-1][..., :self.preds[i - 1][3]]] if i > 0 else []) + [y, skip]
                parts = [F.pad(p, (0, 0, (w - p.shape[3]) // 2, w - p.shape[3] - (w - p.shape[3]) // 2,
                                   (h - p.shape[2]) // 2, h - p.shape[2] - (h - p.shape[2]) // 2)) for p in parts]
                cin = sum(p.shape[-1] for p in parts)
                as_gemm, cp = self._decoder_geometry(i, B, D, h, w, cin)
                cat = torch.zeros((B, D, h, w, cp), dtype=torch.float32, device=y.device) if cp != cin else None
                if cat is None:
                    cat = torch.cat(parts, dim=-1)
                else:
                    torch.cat(parts, dim=-1, out=cat[..., :cin])
                s = self._neuron_bd(cat, sn)                              # MS decoder: SN -> ConvT -> BN
                wkey = "ref"
                self._rec(f"sttmultires_unet.decoders.{i}.sn.spiking_neuron.", s[..., :cin], "BDHWC->TBCHW")
            carried = None
            wuse = self._decoder_weight(i, wkey, y.shape[-1], skip.shape[-1])
            z = torch.empty((B, D, 2 * h, 2 * w, cout), dtype=torch.float32, device=y.device)
            if as_gemm:
                key = ("taps", i, cp, wkey)
                if key not in self._deconv:
                    self._deconv[key] = deconv_tap_weights(wuse, cp, self.nsplit)
                Y = torch.empty((B * D * h * w, 9 * cout), dtype=torch.float32, device=y.device)
                taps = self._deconv[key]
                M1, bc = self._rb(B) * D * h * w, B                  # (rows the routing rules are asked about; samples per launch)
                if getattr(taps, "digits", None) is not None and hip.smallm_gemm_applicable(M1, 9 * cout, cp):
                    taps = taps.digits                        # few rows against many weights (level 0: 1 080 x 3 456 x 1 536): csrc/ms_smallm.hip
                    bc = max(n for n in range(1, B + 1) if B % n == 0 and hip.smallm_gemm_rows_ok(n * D * h * w))
                elif getattr(taps, "digits_rm", None) is not None and hip.res_gemm_applicable(M1, 9 * cout, cp):
                    taps = taps.digits_rm                     # the middle levels (4 320 x 1 728 x 800, 17 280 x 864 x 416): csrc/ms_res.hip
                    bc = max(n for n in range(1, B + 1) if B % n == 0 and hip.res_gemm_applicable(n * D * h * w, 9 * cout, cp))
                if self.replicas and taps is self._deconv[key]:
                    bc = 1            # (16-bit planes on the streaming kernels: their split-K plan follows the row count - one sample per launch)
                mc = bc * D * h * w                                   # (replicas: the kernel family of ONE sample, in sample chunks it admits)
                for m0 in range(0, B * D * h * w, mc):
                    hip.spike_gemm(s.view(-1, cp)[m0:m0 + mc], taps, Y[m0:m0 + mc], mc, 9 * cout, cp)
                hip.deconv_col2im(Y, B * D, h, w, cout, alpha=bn[0], beta=bn[1], out=z)
            # the conv kernel addresses its operands with 31-bit byte offsets: a larger z (config 5: 80 images of
            # 240 x 320 x 96 fp32) goes image chunk by image chunk - the row map of the first n images serves every chunk
            imgs = per = B * D
            fits = lambda n: n * 4 * h * w * cout * 4 < 1 << 31
            if not fits(per):
                whole = [n * D for n in range(B, 0, -1) if B % n == 0 and fits(n * D)]      # whole batch elements, equal chunks
                per = whole[0] if whole else per
            while not fits(per):
                per = (per + 1) // 2
            # the last level: the transposed convolution as ONE digit product over the 2 x 2 input neighbourhood (a row = an input pixel,
            # its 4 cout columns = the 2 x 2 output block; halo tiles in LDS, weights resident: csrc/spike_deconv_wres.hip) instead of
            # four parity-class convolutions
            one_gemm = not as_gemm and self.nsplit == 2 and per % D == 0 and imgs % per == 0 and \
                hip.deconv2x2_applicable(per, D, h, w, cp, cout, fast_only=True)
            if one_gemm:
                key = ("d2x2", i, cp, wkey)
                if key not in self._deconv:
                    self._deconv[key] = (hip.pack_deconv2x2_weight(wuse, cp), bn[0].repeat(4).contiguous(), bn[1].repeat(4).contiguous())
                pl, a4, b4 = self._deconv[key]
                sv, zv = s.view(imgs, h, w, cp), z.view(imgs, 2 * h, 2 * w, cout)
                for i0 in range(0, imgs, per):                            # (image chunks: the kernel's 31-bit offsets, as below)
                    hip.spike_deconv3x3s2(sv[i0:i0 + per], pl, per, D, h, w, cp, cout, alpha=a4, beta=b4, out=zv[i0:i0 + per], tiled_bn=True)
            classes = [] if as_gemm or one_gemm else self._deconv_classes(i, B, D, h, w, cp, wkey, wuse)
            if classes and self.replicas and B > 1:
                # the parity classes run on the streaming kernels (16-bit planes: tile family and split-K plan follow the row count): every
                # sample goes through the launches of its batch-1 forward (the row maps of the first D images serve every sample's slice)
                c1 = self._deconv_classes(i, 1, D, h, w, cp, wkey, wuse)
                sv, zv = s.view(imgs, h, w, cp), z.view(imgs, 4 * h * w, cout)
                for b0 in range(0, imgs, D):
                    hip.spike_conv2d_multi(sv[b0:b0 + D], c1, D, h, w, cp, h, w, zv[b0:b0 + D], alpha=bn[0], beta=bn[1])
                classes = []
            if classes and per == imgs:
                # the four parity classes write disjoint rows of z and each fills about half of the chip: one launch for all of them
                hip.spike_conv2d_multi(s.view(imgs, h, w, cp), classes, imgs, h, w, cp, h, w, z.view(imgs, 4 * h * w, cout), alpha=bn[0], beta=bn[1])
                classes = []
            for cls in classes:
                for i0 in range(0, imgs, per):
                    n = min(per, imgs - i0)
                    hip.spike_conv2d(s.view(imgs, h, w, cp)[i0:i0 + n], cls["Wp"], n, h, w, cp, h, w, cls["KH"], cls["KW"], 1,
                                     cls["dy"], cls["dx"], out=z.view(imgs, 4 * h * w, cout)[i0:i0 + n], alpha=bn[0], beta=bn[1],
                                     out_rowmap=cls["rowmap"][:n * h * w])
            pw, pb, psn, nout, w2, b2 = self.preds[i]
            H2, W2 = 2 * h, 2 * w
            nxt_skip = feats[E - 2 - i] if i + 1 < E else None
            fs = out_size if out_size is not None and out_size[0] % H2 == 0 and out_size[1] % W2 == 0 else None
            if nout == 2 and hip.pred_head_supported(D, cout, fs[0] if fs else H2, fs[1] if fs else W2, H2, W2, psn,
                                                      self.decoders[i + 1][2] if nxt_skip is not None else None) \
                    and B * D * H2 * W2 * max(cout, 16) * 4 < 1 << 40:
                # MS pred: SN -> conv1x1 (+bias), 2 outputs, its time sum + upsampling, and the next level's [y | prediction]
                # spikes - one launch, z read once (csrc/pred_head.hip)
                nxt = None
                if nxt_skip is not None and tuple(nxt_skip.shape[2:4]) == (H2, W2):
                    C1n, C2n = cout, nxt_skip.shape[-1]
                    cin_n = C1n + C2n + 4
                    _, cp_n = self._decoder_geometry(i + 1, B, D, H2, W2, cin_n)
                    carried = ready[i + 1] if ready is not None else torch.empty((B, D, H2, W2, cp_n), dtype=torch.uint8, device=y.device)
                    nxt = (carried, self.decoders[i + 1][2], 0, C1n + C2n, (cin_n, cp_n - cin_n))
                pred, flow, sp = hip.pred_head(z, w2, b2, psn, fs[0] if fs else None, fs[1] if fs else None,
                                               want_pred=self.tape is not None or fs is None or (i + 1 < E and nxt is None),
                                               nxt=nxt, keep=self.tape is not None)
                if sp is not None:
                    self._rec(f"sttmultires_unet.preds.{i}.sn.spiking_neuron.", sp, "BDHWC->TBCHW")
                self._flows[i] = flow
                preds.append(pred)
            else:
                sp = self._neuron_bd(z, psn)                              # MS pred: SN -> conv1x1 (+bias), 2 outputs
                self._rec(f"sttmultires_unet.preds.{i}.sn.spiking_neuron.", sp, "BDHWC->TBCHW")
                po = torch.empty((B * D * 4 * h * w, 32), dtype=torch.float32, device=y.device)
                hip.spike_gemm(sp, pw, po, po.shape[0], 32, cout, bias=pb)
                preds.append(po.view(B, D, 2 * h, 2 * w, 32))             # columns nout.. are exactly zero (zero weight rows, zero bias)
            y = z
        return [None if p is None else p[..., :self.preds[0][3]] for p in preds]

    def forward(self, x, scores=None, replicas=False):
        """(B,bins,2,H,W) fp32 on the GPU -> list of E flow maps (B,2,H,W) (reference :278-305).  `scores` (a list) receives the
        `log=True` output (:283-284): the attention score (T', B_, Wh, Ww, C) of the last block of every stage - what the reference's
        `get_layer_attention_scores` is written to return (its own call chain raises: oracle/sdformer_oracle.py `swin_encoder`).
        `replicas`: the B samples are B INDEPENDENT batch-1 forwards served by one launch sequence - flow i is bit-equal to
        forward(x[i:i+1]) - instead of one reference batch (whose samples window_partition_v2's raw view couples): the
        only difference is the window tables (hip.replica_slice_map / replica_zsrc_map), every kernel sees B times the rows."""
        if not x.is_cuda:
            raise hip.SdfError("input must be a GPU tensor (no CPU fallback)")
        x = x.float().contiguous()
        H, W = x.shape[-2:]
        self.scores = scores
        if replicas and self.num_steps not in (10, 20):
            # other step counts leave layers on the streaming kernels (no digit form at T = 5: the MDR config), whose split-K plans follow the
            # row count - the caller (forward_replicas) serves those models one sample at a time
            raise hip.ReplicaGeometryError(f"replicas are built for T = 10 / 20 (digit kernels throughout), not T = {self.num_steps}")
        self.replicas = bool(replicas)
        try:
            feats = self.encoder(x, **self._tail_kwargs())
            self.scores = None
            preds = self.unet_tail(feats, out_size=(H, W), **({"s1": self.tail_spikes} if getattr(self, "tail_spikes", None) is not None else {}))
        finally:
            self.scores = None
            self.replicas = False
        self.tail_spikes = None
        # sum over time + nearest upsampling to the input size: done by the prediction head's launch, else one small kernel per scale
        return [f if f is not None else hip.flow_out(p, H, W, H / p.shape[2], W / p.shape[3]) for p, f in zip(preds, self._flows)]
