"""Training path of the MS SDformerFlow models (BASELINE config 4): train-mode forward under
autograd, the supervised loss, one optimiser step, and the data-parallel gradient all-reduce.

What is hand-written HIP here: every spiking neuron, forward AND backward (`autograd.LIFFunction` / `PSNFunction` ->
csrc/neuron.hip, csrc/neuron_bwd.hip; 105 neuron calls per forward), the batch-statistics BatchNorm, the token gate, and -
since round 5 - ALL THREE products of every Linear layer (`autograd.LinearHipFunction`: forward and dX on csrc/linear_train.hip,
dW on csrc/linear_dw.hip: 35 ms of library fp32 GEMM -> 13 ms; no rocBLAS GEMM of a Linear layer is left in a step) and the forward and
weight gradient of the MS_ResBlock convolutions (`autograd.Conv3x3HipFunction`: products over zero-ringed channels-last pixel rows).
The other dense products - the convolutions' dX, the remaining convolutions' forward and dW - are library work through torch on the
same stream (MIOpen); their replacements are the next row.  Nothing here touches `oracle/`, and CPU
tensors are refused by the neuron kernels (`SdfError`).

Reference semantics mirrored (file:line under /root/reference):
  model forward           models/STSwinNet_SNN/Spiking_STSwinNet.py:278-305, 161-182
  patch embedding         models/STSwinNet_SNN/Spiking_modules.py:1770-1790 (+ :291-296, :339-347, :811-819, :906-933)
  stage / block / SSA     models/STSwinNet_SNN/Spiking_swin_transformer3D.py:1065-1088, 824-847, 781-821
  QK attention            ...:661-717      window_partition_v2 :100-113    window_reverse swin_transformer3D_v2.py:52-65
  MLP / patch merging     ...:164-181 / :952-974
  BatchNorm in train mode spikingjelly layer.BatchNorm2d, multi-step: flatten (T,B) -> nn.BatchNorm2d (batch statistics)
  DropPath                timm 0.6.13 `drop_path` (per-sample Bernoulli keep mask / keep_prob), SSA branch only (:840)
  loss                    loss/flow_supervised.py:80-105 (gamma None), :14-30
  step                    train_flow_parallel_supervised_SNN.py:233-336 (reset, forward, loss, backward, clip 100, AdamW)
  data parallelism        the reference wraps the model in DataParallel: replicas, local batch statistics, summed
                          gradients; here one process per GPU and ONE bucketed all-reduce(sum)/world per step (RCCL)
"""
import os

import torch
import torch.nn.functional as F

from . import hip


# ---------------------------------------------------------------------------------------------- layers
# Forward of the Linear layers in the training path: 0 = library F.linear (default); 2 / 3 = the inference path's spike GEMM
# with that many weight planes and the activation saved as 1-byte spikes.  Measured at local batch 4: same parity (block-level
# gradients still equal the reference fixture element for element), 2.5 GiB less memory, but the step is SLOWER (fp32 123 ->
# 130 ms: u8 conversion + per-step weight split + fp32 re-expansion in the backward; under bf16 autocast 95 -> 114 ms because
# the backward products leave autocast) - so it stays off until the backward products are spike-aware too (docs/history/DESIGN_rounds1-5.md, round-2 list item 4).
SPIKE_LINEAR_PLANES = 0
# Weight gradient of the Linear layers on csrc/linear_dw.hip (round 5; SDF_TRAIN_LINEAR_DW=0: the library product, for A/B runs)
LINEAR_DW_HIP = os.environ.get("SDF_TRAIN_LINEAR_DW", "1") != "0"
# Forward and dX of the Linear layers on csrc/linear_train.hip (fp32 path; SDF_TRAIN_LINEAR=0: the library products)
LINEAR_HIP = os.environ.get("SDF_TRAIN_LINEAR", "1") != "0"
# ... and of the MS_ResBlock convolutions (3x3 / stride 1 / pad 1 on spikes; SDF_TRAIN_CONV_DW=0: MIOpen's)
CONV_DW_HIP = os.environ.get("SDF_TRAIN_CONV_DW", "1") != "0"
# ... and their forward (SDF_TRAIN_CONV_FWD=0: MIOpen's; the ringed rows are then made in the backward)
CONV_FWD_HIP = os.environ.get("SDF_TRAIN_CONV_FWD", "1") != "0"


def _linear(x, lin, spikes=True):
    """Linear of the training path.  `spikes` (the caller's statement, as `_conv_seq`'s): x holds 0 / 1 values only - every Linear of
    the MS models is fed by a neuron or by the token gate - which is what the hand-written products need: they keep the top 16 bits
    of the activation (csrc/linear_train.hip, linear_dw.hip), exact for spikes and for nothing else.  `spikes=False`: F.linear.
    SDF_DEBUG_CHECKS=1 verifies the statement on the host (one synchronisation per call)."""
    if not spikes:
        return F.linear(x, lin.weight, lin.bias)
    if hip.sw("SDF_DEBUG_CHECKS", "") == "1" and not bool(((x == 0) | (x == 1)).all()):
        raise hip.SdfError("_linear(spikes=True) was handed values other than 0 / 1: the 16-bit-plane products would truncate them")
    K, N = lin.weight.shape[1], lin.weight.shape[0]
    if SPIKE_LINEAR_PLANES and K % 32 == 0 and N % 32 == 0:
        from .autograd import SpikeLinearFunction
        return SpikeLinearFunction.apply(x, lin.weight, lin.bias, SPIKE_LINEAR_PLANES)
    if LINEAR_DW_HIP and x.is_cuda and hip.linear_dw_applicable(x.numel() // K, N, K):
        from .autograd import LinearDwFunction, LinearHipFunction
        if LINEAR_HIP and not torch.is_autocast_enabled():               # (bf16 autocast keeps the library's bf16 forward / dX)
            return LinearHipFunction.apply(x, lin.weight, lin.bias)
        return LinearDwFunction.apply(x, lin.weight, lin.bias)
    return F.linear(x, lin.weight, lin.bias)


def _bn(x_nc, bn):
    """Batch-statistics BatchNorm over dim 1 of (N, C, H, W) with the module's parameters; running stats updated in place."""
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    if hip.bn_train_nchw_supported(x_nc):
        from .autograd import BatchNormNCHWFunction
        return BatchNormNCHWFunction.apply(x_nc, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)
    return F.batch_norm(x_nc, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)   # H*W % 4 != 0


def _bn_last(x, bn):
    """(T, B, ..., C) channel-last: the reference permutes channels to dim 2, flattens (T, B) and applies BatchNorm2d - the
    same statistics are the column statistics of the (rows, C) matrix, taken in place by the HIP kernels (no permute copies)."""
    from .autograd import BatchNormLastFunction
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    return BatchNormLastFunction.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps)


def _bn_ch2(x, bn):
    """(T, B, C, H, W) -> BatchNorm2d on the (T*B, C, H, W) view."""
    return _bn(x.flatten(0, 1), bn).view(x.shape)


def _conv_seq(x, conv, stride=None, padding=None, spikes=False):
    """layer.Conv2d in multi-step mode: (T, B, C, H, W) through one conv2d call on the flattened batch.  spikes=True: x is a neuron's
    output - a 3x3 / stride 1 / pad 1 convolution with 96-multiples of channels then takes its WEIGHT gradient on csrc/linear_dw.hip
    (convolution form); forward and dX stay library convolutions."""
    T, B = x.shape[:2]
    x2 = x.flatten(0, 1)
    if (spikes and CONV_DW_HIP and stride is None and padding is None and x2.is_cuda and tuple(conv.kernel_size) == (3, 3)
            and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and hip.conv3x3_dw_applicable(x2.shape[0], conv.in_channels, conv.out_channels, x2.shape[2], x2.shape[3])):
        from .autograd import Conv3x3DwFunction, Conv3x3HipFunction
        if CONV_FWD_HIP and not torch.is_autocast_enabled() and conv.in_channels % 32 == 0:
            y = Conv3x3HipFunction.apply(x2, conv.weight, conv.bias)
        else:
            y = Conv3x3DwFunction.apply(x2, conv.weight, conv.bias)
    else:
        y = F.conv2d(x2, conv.weight, conv.bias, conv.stride if stride is None else stride, conv.padding if padding is None else padding)
    return y.view(T, B, *y.shape[1:])


def _drop_path(x, p, training):
    if p <= 0.0 or not training:
        return x
    keep = 1.0 - p
    mask = torch.empty((x.shape[0],) + (1,) * (x.dim() - 1), dtype=x.dtype, device=x.device).bernoulli_(keep)
    return x * (mask / keep)


def rebin_events(x, num_bins, num_steps):
    """(B, bins, 2, H, W) -> (T, B, num_ch, H, W): time step t takes bin t of each polarity (patch embedding :1775-1784)."""
    if x.size(1) > num_bins:
        x = x[:, :num_bins]
    num_ch = num_bins * 2 // num_steps
    ev = x.permute(0, 2, 3, 4, 1)
    chans = [ev[:, i % 2, :, :, (i // 2) * num_steps:(i // 2 + 1) * num_steps] for i in range(num_ch)]
    return torch.stack(chans, 1).permute(4, 0, 1, 2, 3).contiguous()


def ms_resblock(x, rb):
    y = rb.sn1(x)
    y = _bn_ch2(_conv_seq(y, rb.conv1[0], spikes=True), rb.norm1.norm_layer)
    y = rb.sn2(y)
    y = _bn_ch2(_conv_seq(y, rb.conv2[0], spikes=True), rb.norm2.norm_layer)
    return y + x


def patch_embed(x, pe):
    x = rebin_events(x, pe.num_bins, pe.num_steps)
    y = pe.head.sn(_bn_ch2(_conv_seq(x, pe.head.conv[0]), pe.head.norm_layer.norm_layer))
    y = _bn_ch2(_conv_seq(y, pe.conv.conv[0]), pe.conv.norm_layer.norm_layer)
    for rb in pe.residual_encoding.resblocks:
        y = ms_resblock(y, rb)
    T, B = y.shape[:2]
    res = F.conv2d(y.flatten(0, 1), pe.proj.conv_res.weight, pe.proj.conv_res.bias, stride=2)
    s = pe.proj.sn(y)
    z = F.conv2d(s.flatten(0, 1), pe.proj.conv.weight, pe.proj.conv.bias, stride=2, padding=1)
    z = _bn(z, pe.proj.norm_layer) + res
    return z.view(T, B, *z.shape[1:])


def get_window_size(x_size, window_size, shift_size):
    ws, ss = list(window_size), list(shift_size)
    for i in range(3):
        if x_size[i] <= window_size[i]:
            ws[i], ss[i] = x_size[i], 0
    return tuple(ws), tuple(ss)


def qk_attention(x, attn):
    """x (T', B_, N1, C): slice list of the windows, T' = window depth.  Returns (T', B_, N1, C)."""
    Tq, B_, N1, C = x.shape
    nH = attn.num_heads
    hd = C // nH
    xs = attn.proj_sn(x)
    q = attn.sn_q(_bn_last(_linear(xs, attn.linear_q), attn.bn_q.norm_layer))
    k = _bn_last(_linear(xs, attn.linear_k), attn.bn_k.norm_layer)
    k = attn.sn_k(k + attn.positional_encoding.reshape(Tq, 1, N1, C))
    gate = attn.sn2_q.spiking_neuron
    if hd == 32 and Tq in (1, 2, 4):                                    # token gate, forward and backward one HIP launch each
        from .autograd import QKGateFunction
        alpha = getattr(gate.surrogate_function, "alpha", 2.0)
        if gate.kind == "psn":
            e = QKGateFunction.apply(q, k, gate.weight, gate.bias, "psn", 2.0, 0.0, None, True, alpha)
        else:
            e = QKGateFunction.apply(q, k, None, None, gate.kind, gate.tau, gate.v_threshold, gate.v_reset, gate.detach_reset, alpha)
    else:                                                               # other head widths / window depths: composed expression
        a = attn.sn2_q(q.reshape(Tq, B_, N1, nH, hd).sum(-1))
        e = k * a.repeat_interleave(hd, dim=-1)
    z = e.reshape(B_, nH, Tq, N1, hd).permute(2, 0, 3, 1, 4).reshape(Tq, B_, N1, C)     # the reference's raw head reshape
    return _bn_last(_linear(z, attn.proj), attn.proj_bn.norm_layer)


_SLICE_MAPS = {}


def ssa(x, blk):
    """Window partition -> attention -> window reverse, both as row moves through the int32 slice map (built on the device,
    cached per shape): no materialised pad / roll / permute / crop in either direction of autograd."""
    from .autograd import WindowGatherFunction, WindowScatterFunction
    B, D, H, W, C = x.shape
    (Wd, Wh, Ww), ss = get_window_size((D, H, W), blk.window_size, blk.shift_size)
    key = (B, D, H, W, Wd, Wh, Ww, ss, str(x.device))
    if key not in _SLICE_MAPS:
        _SLICE_MAPS[key] = hip.window_slice_map(B, D, H, W, (Wd, Wh, Ww), ss, x.device)
    row_map, B_ = _SLICE_MAPS[key]
    xw = WindowGatherFunction.apply(x, row_map)                          # (Wd * B_ * Wh * Ww, C): step t' of window b' is slice t' B_ + b'
    y = qk_attention(xw.view(Wd, B_, Wh * Ww, C), blk.attn)
    return WindowScatterFunction.apply(y.reshape(-1, C), row_map, (B, D, H, W, C))


def ms_mlp(x, mlp):
    h = _bn_last(_linear(mlp.sn1(x), mlp.fc1), mlp.bn1.norm_layer)
    return _bn_last(_linear(mlp.sn2(h), mlp.fc2), mlp.bn2.norm_layer)


def ms_block(x, blk, training=True):
    x = _drop_path(ssa(x, blk), blk.drop_path_rate, training) + x
    return ms_mlp(x.permute(1, 0, 2, 3, 4).contiguous(), blk.mlp).permute(1, 0, 2, 3, 4) + x


def ms_patch_merge(x, pm):
    B, D, H, W, C = x.shape
    if H % 2 or W % 2:
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, :, 0::2, 0::2], x[:, :, 1::2, 0::2], x[:, :, 0::2, 1::2], x[:, :, 1::2, 1::2]], -1)
    x = pm.sn(x.permute(1, 0, 2, 3, 4).contiguous())
    return _bn_last(_linear(x, pm.reduction), pm.norm.norm_layer).permute(1, 0, 2, 3, 4)


def skip_concat_ch(x1, x2):
    dY, dX = x2.shape[-2] - x1.shape[-2], x2.shape[-1] - x1.shape[-1]
    return torch.cat([F.pad(x1, (dX // 2, dX - dX // 2, dY // 2, dY - dY // 2)), x2], dim=2)


def forward_train(model, x):
    """(B, bins, 2, H, W) on the GPU -> list of flow maps (B, 2, H, W), differentiable; BN running stats are updated."""
    if not x.is_cuda:
        raise hip.SdfError("the training path runs on the GPU only (no CPU fallback)")
    H, W = x.shape[-2:]
    unet = model.sttmultires_unet
    sw = unet.encoders.swin3d
    y = patch_embed(x.float(), sw.patch_embed).permute(1, 0, 3, 4, 2).contiguous()          # (B, D, h, w, C)
    blocks = []
    for i, layer in enumerate(sw.layers):
        for blk in layer.swin_blocks:
            y = ms_block(y, blk, model.training)
        if i in sw.out_indices:
            blocks.append(y.permute(1, 0, 4, 2, 3).contiguous())                            # (D, B, C, h, w)
        if layer.downsample is not None:
            y = ms_patch_merge(y, layer.downsample)
    y = blocks[-1]
    for rb in unet.resblocks:
        y = ms_resblock(y, rb)
    preds, E = [], len(blocks)
    for i in range(E):
        y = skip_concat_ch(y, blocks[E - i - 1])
        if i > 0:
            y = skip_concat_ch(preds[-1], y)
        dec, pr = unet.decoders[i], unet.preds[i]
        s = dec.sn(y.contiguous())
        T, B = s.shape[:2]
        dc = dec.deconv[0]
        z = F.conv_transpose2d(s.flatten(0, 1), dc.weight, dc.bias, stride=2, padding=dc.kernel_size[0] // 2, output_padding=1)
        y = _bn_ch2(z.view(T, B, *z.shape[1:]), dec.norm_layer.norm_layer)
        preds.append(_conv_seq(pr.sn(y), pr.conv[0], padding=0))
    flows = []
    for f in preds:
        f = f.sum(0)
        flows.append(F.interpolate(f, scale_factor=(H / f.shape[-2], W / f.shape[-1])))
    return flows


# ---------------------------------------------------------------------------------------------- loss, step, all-reduce
def flow_loss_supervised(pred_list, gt_flow, mask, flow_scaling=1.0, lambda_mod=1.0, n_valid=None):
    """Mean over predictions and samples of sum(EPE * mask) / (number of valid pixels) (reference loss/flow_supervised.py:85-102).
    `n_valid` replaces the local count when the batch is sharded over ranks (see `global_valid_count`)."""
    n = torch.sum(mask) if n_valid is None else n_valid
    cur = 0.0
    for pred in pred_list:
        flow = pred * flow_scaling
        err = torch.sqrt((flow - gt_flow).pow(2).sum(1) + 1e-8).view(flow.shape[0], -1) * mask.reshape(flow.shape[0], -1)
        cur = cur + lambda_mod * (torch.sum(err, dim=1) / (n + 1e-9))
    return torch.mean(cur / len(pred_list))


def global_valid_count(mask, dist=None, world=1):
    """Valid pixels of the GLOBAL batch.  The reference trains under nn.DataParallel: the replicas' outputs are gathered and
    the loss is taken on the whole batch, so every sample's error is divided by the valid-pixel count of the whole batch
    (loss/flow_supervised.py:90, train_flow_parallel_supervised_SNN.py:139-143, :306-308).  With one process per GPU the
    count is summed over the ranks (one scalar all-reduce, issued before the backward pass); together with the mean over
    ranks of the gradients this reproduces the gathered-batch gradient exactly - normalising by the local count instead
    makes the gradient ~world times larger."""
    n = torch.sum(mask).float()
    if dist is not None and world > 1:
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return n


class GradientBuckets:
    """Flat fp32 gradient buckets for the data-parallel step: the parameters' `.grad` are VIEWS into a few large
    contiguous buffers (default 64 MiB - ring all-reduce over xGMI is per-link bound, few large messages beat many small
    ones; the whole en4 model is 220 MB = 4 buckets), so a bucket's all-reduce is one collective with no packing copies.

    Overlap with backward: every parameter carries a post-accumulate-grad hook; when the last parameter of a bucket has
    its gradient the bucket's all-reduce is launched asynchronously right there, inside `loss.backward()` (buckets are
    filled in reverse parameter order = the order backward produces gradients, so the first collective leaves while most
    of the backward pass is still to run).  `finish()` launches whatever is left - buckets holding a parameter that
    received no gradient this step - waits for all, divides by world, and hands parameters that never received a
    gradient to the optimiser as `.grad = None` (the reference leaves them None: no weight decay / moment updates on
    the dead attn_sn weights).  Every rank launches the same collectives in the same order: completion order is a
    property of the autograd graph, the left-overs go in bucket order."""

    def __init__(self, params, bucket_bytes=64 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.buckets, cur, cur_n = [], [], 0
        for p in reversed(self.params):                                # backward produces gradients last layer first
            if cur and (cur_n + p.numel()) * 4 > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat, self._views, self._bucket_of = [], {}, {}
        for bi, b in enumerate(self.buckets):
            buf = torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
            off = 0
            for p in b:
                self._views[p] = buf[off:off + p.numel()].view_as(p)
                self._bucket_of[p] = bi
                p.grad = self._views[p]
                off += p.numel()
                p.register_post_accumulate_grad_hook(self._ready)
            self.flat.append(buf)
        self._dist, self._world = None, 1
        self._pending = [len(b) for b in self.buckets]
        self._seen, self._works, self._launched = set(), [], set()
        self.log = []                                                   # (event, bucket index, hooks fired so far): test hook

    def zero(self):
        for buf in self.flat:
            buf.zero_()

    def begin(self, dist=None, world=1):
        """Start of a step: zero the buckets, re-attach the `.grad` views, arm the hooks."""
        self.zero()
        for p, v in self._views.items():
            p.grad = v
        self._dist, self._world = (dist, world) if (dist is not None and world > 1) else (None, 1)
        self._pending = [len(b) for b in self.buckets]
        self._seen, self._works, self._launched, self.log = set(), [], set(), []

    def _launch(self, bi):
        self._launched.add(bi)
        self.log.append(("all_reduce", bi, len(self._seen)))
        if self._dist is not None:
            self._works.append(self._dist.all_reduce(self.flat[bi], op=self._dist.ReduceOp.SUM, async_op=True))

    def _ready(self, p):
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        if p.grad is not self._views[p]:                                # autograd replaced the view (first accumulation into None)
            self._views[p].copy_(p.grad)
            p.grad = self._views[p]
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            self._launch(bi)

    def finish(self):
        """After backward: remaining buckets, wait, average, `.grad = None` for parameters without a gradient."""
        for bi in range(len(self.buckets)):
            if bi not in self._launched:
                self._launch(bi)
        for w in self._works:
            w.wait()
        if self._dist is not None:
            for buf in self.flat:
                buf.div_(self._world)
        for p in self.params:
            if id(p) not in self._seen:
                p.grad = None
        self._works = []

    def all_reduce(self, dist=None, world=1):
        """Non-overlapped form (gradients already complete): sum over ranks / world, one collective per bucket."""
        if dist is None or world <= 1:
            return
        works = [dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True) for buf in self.flat]
        for w in works:
            w.wait()
        for buf in self.flat:
            buf.div_(world)


def train_step(model, optimizer, chunk, label, mask, buckets=None, dist=None, world=1, clip_grad=100.0, flow_scaling=1.0,
               lambda_mod=1.0, amp=False, forward_fn=None):
    """One step of train_flow_parallel_supervised_SNN.py's loop body (:233-336) on this rank's shard of the batch; returns the
    loss tensor (this rank's samples' mean of err / n_global; the mean over ranks is the gathered-batch loss).
    `amp`: the reference trains under `torch.cuda.amp.autocast` with fp16 + GradScaler (`optimizer.use_amp: true`,
    :248, :314-331); here the same regions run under bf16 autocast (no scaler needed) - spikes are exact in bf16, the
    membranes / neuron kernels, BatchNorm statistics, the loss and the optimiser stay fp32.
    `forward_fn(model, chunk) -> flows` replaces the HIP train-mode forward (the CPU dry run of the N > 1 plumbing, bench.py --train
    --plumbing: everything else in this function is the code the GPU ranks run)."""
    from .spikingjelly_compat import functional
    model.train()
    functional.reset_net(model)
    if buckets is not None:
        buckets.begin(dist, world)
    else:
        optimizer.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
        flows = forward_train(model, chunk) if forward_fn is None else forward_fn(model, chunk)
    n_valid = global_valid_count(mask, dist, world)
    # local mean over B_local of err_i / n_global, times 1 / world from the gradient average = the gathered-batch mean
    loss = flow_loss_supervised([f.float() for f in flows], label, mask, flow_scaling, lambda_mod, n_valid=n_valid)
    loss.backward()                                                  # bucket all-reduces leave from the grad-ready hooks
    if buckets is not None:
        buckets.finish()
    if clip_grad is not None:
        torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], clip_grad)
    optimizer.step()
    return loss.detach()
