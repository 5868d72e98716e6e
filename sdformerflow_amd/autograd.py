"""torch.autograd bridges for the training path: forward and backward are both HIP kernels
(csrc/neuron.hip, csrc/neuron_bwd.hip); nothing is saved between them except the input itself - the backward
recomputes the membrane trajectory.  No CPU fallback: CPU tensors raise `SdfError` inside `hip`."""
import os

import torch

from . import hip

_LIBRARY_DX = os.environ.get("SDF_TRAIN_LINEAR_DX", "1") == "0"


class LIFFunction(torch.autograd.Function):
    """Multi-step LIF / IF over dim 0: spikes = LIF(x); dL/dx by BPTT with the ATan surrogate
    (reference: autograd through spikingjelly's LIFNode as built in Spiking_modules.py:40-66)."""

    @staticmethod
    def forward(ctx, x, tau, v_th, v_reset, detach_reset, alpha, kind):
        ctx.in_dtype = x.dtype
        x = x.float().contiguous()                  # under bf16 autocast the producer hands over bf16: the membrane is fp32
        ctx.save_for_backward(x)
        ctx.cfg = (tau, v_th, v_reset, detach_reset, alpha, kind)
        if kind == "if":
            raise hip.SdfError("IFNode has no forward call site in the reference (Spiking_STSwinNet.py:154-156)")
        return hip.lif_fwd(x, tau, v_th, v_reset, torch.float32)

    @staticmethod
    def backward(ctx, grad_spike):
        (x,) = ctx.saved_tensors
        tau, v_th, v_reset, detach_reset, alpha, kind = ctx.cfg
        gx = hip.lif_bwd(x, grad_spike.float(), tau, v_th, v_reset, detach_reset, alpha, kind)
        return gx.to(ctx.in_dtype), None, None, None, None, None, None


class PSNFunction(torch.autograd.Function):
    """Parallel spiking neuron over dim 0: spikes = (b + W x >= 0); (dL/dx, dL/dW, dL/db) with the ATan surrogate
    (reference PSN.forward, Spiking_submodules.py:207-211)."""

    @staticmethod
    def forward(ctx, x, W, b, alpha):
        ctx.in_dtype = x.dtype
        x = x.float().contiguous()
        ctx.save_for_backward(x, W, b)
        ctx.alpha = alpha
        return hip.psn_fwd(x, W.detach().float(), b.detach().float(), torch.float32)

    @staticmethod
    def backward(ctx, grad_spike):
        x, W, b = ctx.saved_tensors
        need = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        gx, gW, gb = hip.psn_bwd(x, W.detach().float(), b.detach().float(), grad_spike.float(), ctx.alpha, need_param_grads=need)
        return gx.to(ctx.in_dtype), gW, (gb.view_as(b) if gb is not None else None), None


class QKGateFunction(torch.autograd.Function):
    """Token gate of the QK attention, e = k * SN2_q(sum of q over each head's channels), both directions one HIP launch
    (reference Spiking_swin_transformer3D.py:687-694 under autograd).  `psn_w` / `psn_b` are the gate's PSN parameters
    (None for LIF / IF): their gradients come out of the same backward launch."""

    @staticmethod
    def forward(ctx, q, k, psn_w, psn_b, kind, tau, v_th, v_reset, detach_reset, alpha):
        ctx.in_dtype = q.dtype
        q, k = q.float().contiguous(), k.float().contiguous()
        p = hip.NeuronParams(kind, tau, v_th, v_reset,
                             psn_w=None if psn_w is None else psn_w.detach().float().contiguous(),
                             psn_b=None if psn_b is None else psn_b.detach().float().reshape(-1).contiguous())
        ctx.save_for_backward(q, k)
        ctx.cfg = (p, detach_reset, alpha, None if psn_b is None else psn_b.shape)
        return hip.qk_gate_f32(q, k, p)

    @staticmethod
    def backward(ctx, grad_e):
        q, k = ctx.saved_tensors
        p, detach_reset, alpha, bshape = ctx.cfg
        gq, gk, gW, gb = hip.qk_gate_bwd(q, k, grad_e.float(), p, detach_reset, alpha)
        return (gq.to(ctx.in_dtype), gk.to(ctx.in_dtype), gW, None if gb is None else gb.view(bshape),
                None, None, None, None, None, None)


class BatchNormLastFunction(torch.autograd.Function):
    """Batch-statistics BatchNorm over the last dim of a channel-last tensor (sdf_bn_train_fwd / _bwd): the training form of
    the reference's `SpikingNormLayer("BN")(x.permute(0,1,4,2,3)).permute(...)` without the permute copies."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        ctx.in_dtype = x.dtype
        x2 = x.float().contiguous().view(-1, x.shape[-1])
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y, mean, invstd = hip.bn_train_fwd(x2, w, b, running_mean, running_var, momentum, eps)
        ctx.save_for_backward(x2, w, mean, invstd)
        ctx.shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, gy):
        x2, w, mean, invstd = ctx.saved_tensors
        gx, gw, gb = hip.bn_train_bwd(x2, gy.float().contiguous().view(x2.shape), w, mean, invstd)
        return gx.view(ctx.shape).to(ctx.in_dtype), gw, gb, None, None, None, None


class BatchNormNCHWFunction(torch.autograd.Function):
    """Batch-statistics BatchNorm2d on an (N, C, H, W) tensor (sdf_bn_train_nchw_fwd / _bwd): the conv outputs of the patch
    embedding and the U-Net tail in training mode."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps):
        ctx.in_dtype = x.dtype
        x4 = x.float().contiguous()
        w, b = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y, mean, invstd = hip.bn_train_nchw_fwd(x4, w, b, running_mean, running_var, momentum, eps)
        ctx.save_for_backward(x4, w, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, gy):
        x4, w, mean, invstd = ctx.saved_tensors
        gx, gw, gb = hip.bn_train_nchw_bwd(x4, gy.float().contiguous(), w, mean, invstd)
        return gx.to(ctx.in_dtype), gw, gb, None, None, None, None


class WindowGatherFunction(torch.autograd.Function):
    """Window partition as a row gather through the slice map (pad + roll(-shift) + window_partition_v2 of the reference,
    Spiking_swin_transformer3D.py:789-804, :100-113) - backward is the matching scatter."""

    @staticmethod
    def forward(ctx, x, row_map):
        ctx.in_dtype, ctx.shape = x.dtype, x.shape
        ctx.save_for_backward(row_map)
        return hip.rows_gather(x.float().contiguous().view(-1, x.shape[-1]), row_map)

    @staticmethod
    def backward(ctx, g):
        (row_map,) = ctx.saved_tensors
        rows = 1
        for d in ctx.shape[:-1]:
            rows *= d
        return hip.rows_scatter(g.float().contiguous(), row_map, rows).view(ctx.shape).to(ctx.in_dtype), None


class WindowScatterFunction(torch.autograd.Function):
    """Window reverse as a row scatter through the same map (window_reverse + roll(+shift) + crop, :810-820) - backward is the gather."""

    @staticmethod
    def forward(ctx, y2, row_map, out_shape):
        ctx.in_dtype = y2.dtype
        ctx.save_for_backward(row_map)
        rows = 1
        for d in out_shape[:-1]:
            rows *= d
        return hip.rows_scatter(y2.float().contiguous(), row_map, rows).view(out_shape)

    @staticmethod
    def backward(ctx, g):
        (row_map,) = ctx.saved_tensors
        return hip.rows_gather(g.float().contiguous().view(-1, g.shape[-1]), row_map).to(ctx.in_dtype), None, None


class LinearDwFunction(torch.autograd.Function):
    """nn.Linear on a SPIKE tensor with its weight gradient on csrc/linear_dw.hip: dW = dY^T X has a reduction as long as the
    token count (276 480 at stage 0, local batch 4) and a 96 x 96 ... 768 x 3 072 result - the library's fp32 product took 20 ms of a
    117 ms step on it, the kernel 4 ms (tools/linear_dw_bench.py).  Forward and dX = dY W stay library products.  The activation
    is saved as autograd would save it (fp32); what the kernel needs of it is that its values are exact in bf16 (0 / 1).
    Reference: autograd through nn.Linear in train_flow_parallel_supervised_SNN.py:233-336."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        N, K = weight.shape
        g2 = g.reshape(-1, N)
        gx = (g2 @ weight.to(g2.dtype)).view(x.shape).to(x.dtype) if ctx.needs_input_grad[0] else None
        gw = None
        if ctx.needs_input_grad[1]:
            gw = hip.linear_dw(g2.float().contiguous(), x.reshape(-1, K).float().contiguous()).to(weight.dtype)
        gb = g2.sum(0).to(weight.dtype) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


class LinearHipFunction(torch.autograd.Function):
    """nn.Linear on a SPIKE tensor with all three of its products on hand-written kernels: forward and dX = dY W on
    csrc/linear_train.hip (operands split into exact bf16 planes inside the kernel, straight from the fp32 weight - nothing to pack,
    nothing to keep in step with the optimiser), dW = dY^T X on csrc/linear_dw.hip.  fp32 in, fp32 out.
    Reference: nn.Linear forward / autograd in train_flow_parallel_supervised_SNN.py:233-336."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        K = weight.shape[1]
        x2 = x.reshape(-1, K).float().contiguous()
        ctx.save_for_backward(x2, weight)
        ctx.has_bias, ctx.xshape, ctx.in_dtype = bias is not None, x.shape, x.dtype
        w = weight.detach().float().contiguous()
        out = hip.linear_train(x2, w, None if bias is None else bias.detach().float().contiguous(), mode=0)
        return out.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, weight = ctx.saved_tensors
        g2 = g.reshape(-1, weight.shape[0]).float().contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if _LIBRARY_DX:                                                  # diagnostic (SDF_TRAIN_LINEAR_DX=0): the library's dX behind the HIP forward
                gx = (g2 @ weight.detach().float()).view(ctx.xshape).to(ctx.in_dtype)
            else:
                gx = hip.linear_train(g2, weight.detach().float().contiguous(), mode=1).view(ctx.xshape).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            gw = hip.linear_dw(g2, x2).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0).to(weight.dtype)
        return gx, gw, gb


class Conv3x3DwFunction(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 Conv2d on a SPIKE image (the MS_ResBlock convolutions of the patch embedding and the bottleneck,
    reference Spiking_modules.py:291-347) with its weight gradient on csrc/linear_dw.hip's convolution form: 1.1 M pixels x 96 x 864
    per layer of the patch embedding at local batch 4.  Forward and dX are library convolutions, as before."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.conv2d(x, weight, bias, 1, 1)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.nn.grad.conv2d_input(x.shape, weight.to(g.dtype), g, stride=1, padding=1).to(x.dtype)
        if ctx.needs_input_grad[1]:
            gw = hip.conv3x3_dw(g.float().contiguous(), x.float()).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum((0, 2, 3)).to(weight.dtype)
        return gx, gw, gb


class Conv3x3HipFunction(torch.autograd.Function):
    """3x3 / stride 1 / pad 1 Conv2d on a SPIKE image with its FORWARD and its weight gradient on hand-written kernels: the image goes
    to zero-ringed channels-last pixel rows once (csrc/linear_dw.hip `ringed_rows_kernel`), the forward is a product over those rows
    with the tap as a row offset of the loader (csrc/linear_train.hip, convolution form; spike plane x three exact bf16 weight
    planes), and the rows are what the backward's dW product needs - they are saved instead of the image.  dX stays MIOpen's
    (a product of real-valued dY with eight plane pairs would not beat its Winograd kernel).
    Reference: MS_ResBlock's convolutions, Spiking_modules.py:291-347."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        imgs, Cc, H, W = x.shape
        xr = hip.ringed_rows(x.float().contiguous())
        ctx.save_for_backward(xr, weight)
        ctx.has_bias, ctx.xshape, ctx.in_dtype = bias is not None, tuple(x.shape), x.dtype
        return hip.conv3x3_fwd_ringed(xr, weight, None if bias is None else bias.detach().float().contiguous(), imgs, H, W)

    @staticmethod
    def backward(ctx, g):
        xr, weight = ctx.saved_tensors
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = torch.nn.grad.conv2d_input(ctx.xshape, weight.to(g.dtype), g, stride=1, padding=1).to(ctx.in_dtype)
        if ctx.needs_input_grad[1]:
            gw = hip.conv3x3_dw(g.float().contiguous(), xr).to(weight.dtype)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g.sum((0, 2, 3)).to(weight.dtype)
        return gx, gw, gb


class SpikeLinearFunction(torch.autograd.Function):
    """Linear layer on a spike tensor in the training path: the FORWARD is the inference path's spike GEMM (binary activations
    exact in 16 bits, fp32-grade weight planes re-split from the current weights, fp32 accumulate) and the activation is kept
    for the backward as 1-byte spikes (a quarter of what autograd would hold); the backward's two dense products
    (dX = dY W, dW = dY^T X) are library GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, nsplit):
        K, N = x.shape[-1], weight.shape[0]
        xu8 = x.reshape(-1, K).to(torch.uint8)
        planes = hip.split_weight(weight.detach().float().contiguous(), nsplit)
        out = torch.empty((xu8.shape[0], N), dtype=torch.float32, device=x.device)
        hip.spike_gemm(xu8, planes, out, xu8.shape[0], N, K, bias=None if bias is None else bias.detach().float().contiguous())
        ctx.save_for_backward(xu8, weight)
        ctx.has_bias, ctx.xshape, ctx.in_dtype = bias is not None, x.shape, x.dtype
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, g):
        xu8, weight = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1]).float()
        gx = (g2 @ weight.float()).view(ctx.xshape).to(ctx.in_dtype) if ctx.needs_input_grad[0] else None
        gw = g2.t() @ xu8.float() if ctx.needs_input_grad[1] else None
        gb = g2.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb, None
