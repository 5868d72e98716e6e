"""`forward()` of the block-level classes of the module tree (reference models/STSwinNet_SNN/Spiking_swin_transformer3D.py
:661-717, :164-181, :147-162, :824-847, :952-974, :914-934).

The whole-model forward goes through `engine.MSFlowEngine` / `engine_sew.SEWFlowEngine`; a user who instantiates ONE block
(the reference's own smoke blocks do) gets the same kernels here: the module packs its weights once
(cached on the parameters' version counters) and runs through a model-less engine (`MSFlowEngine.bare`).  Tensors cross
the module boundary in the reference's layouts; inside they are the engine's channel-last buffers.  HIP only: CPU tensors raise.
"""
from __future__ import annotations

import torch

from . import hip


def _stamp(mod):
    return tuple(t._version for t in list(mod.parameters()) + list(mod.buffers()))


def packed(mod, build):
    """Per-module cache of its packed form (weight planes, folded BN), rebuilt when a parameter changed or moved."""
    dev = next(mod.parameters()).device
    key = (str(dev), _stamp(mod))
    if getattr(mod, "_sdf_packed_key", None) != key:
        mod._sdf_packed, mod._sdf_packed_key = build(dev), key
    return mod._sdf_packed


def _eval_only(mod):
    if mod.training:
        raise NotImplementedError(f"{type(mod).__name__}.forward: inference only at module level (training goes through the model: "
                                  "sdformerflow_amd.train)")


def _cl(x):
    if not x.is_cuda:
        raise hip.SdfError("module forwards need device tensors (no CPU fallback)")
    return x.float().contiguous()


# ------------------------------------------------------------------------------------------------ MS family
def ms_block_forward(mod, x, mask_matrix=None, return_attention=False):
    """`MS_Spiking_SwinTransformerBlock3D.forward` (reference :824-847): x (B,D,H,W,C) -> same shape.  The QK attention
    takes no mask (:661), `mask_matrix` is accepted and unused like in the reference."""
    from .engine import MSFlowEngine, _Block
    _eval_only(mod)
    eng, blk = packed(mod, lambda dev: (MSFlowEngine.bare(dev), _Block(mod, dev, 2)))
    with torch.no_grad():
        y = _cl(x).clone()                                             # the engine updates the stream in place
        if return_attention:                                           # :836-837: the score (T',B_,Wh,Ww,C) instead of the block's output
            score = []
            eng.attention(y, blk, score)
            return score[0]
        return eng.mlp(eng.attention(y, blk), blk)


def ms_mlp_forward(mod, x):
    """`MS_Spiking_Mlp.forward` (reference :164-181): x (T,B,H,W,C) -> BN2(fc2(SN2(BN1(fc1(SN1(x)))))), no shortcut."""
    from .engine import _Lin, _np
    _eval_only(mod)
    fc1, fc2, sn1, sn2 = packed(mod, lambda dev: (_Lin(mod.fc1, mod.bn1.norm_layer, dev, 2), _Lin(mod.fc2, mod.bn2.norm_layer, dev, 2),
                                                  _np(mod.sn1, dev), _np(mod.sn2, dev)))
    with torch.no_grad():
        x = _cl(x)
        T, n = x.shape[0], x[0].numel()
        rows = n // fc1.K
        s1 = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        hip.neuron_fwd(x, s1, T, 1, n, 0, n, 0, n, sn1)
        s2 = torch.empty((T * rows, fc1.N), dtype=torch.uint8, device=x.device)
        hip.spike_gemm_sn(s1, fc1.Wp, s2, fc1.N, fc1.K, T, rows, rows, 0, rows, sn2, alpha=fc1.alpha, beta=fc1.beta)
        out = torch.empty(x.shape[:-1] + (fc2.N,), dtype=torch.float32, device=x.device)
        hip.spike_gemm(s2, fc2.Wp, out, T * rows, fc2.N, fc2.K, alpha=fc2.alpha, beta=fc2.beta)
        return out


def qk_attention_forward(mod, x, mask=None):
    """`Spiking_QK_WindowAttention3D.forward` (reference :661-717): x (T',B_,Wh,Ww,C) window slices -> (out (B_, T'*Wh*Ww, C),
    attn (T',B_,Wh,Ww,C)), attn = `attn_sn(Z)` (:711) - the attention score that `log=True` collects."""
    from .engine import _Block
    _eval_only(mod)

    def build(dev):
        from .engine import _Lin, _np
        b = _Block.__new__(_Block)
        a = mod
        b.name, b.nH = "", a.num_heads
        b.q, b.k = _Lin(a.linear_q, a.bn_q.norm_layer, dev, 2), _Lin(a.linear_k, a.bn_k.norm_layer, dev, 2)
        b.p = _Lin(a.proj, a.proj_bn.norm_layer, dev, 2)
        b.pe = a.positional_encoding.detach().float().to(dev).contiguous()
        b.sn_proj, b.sn_q, b.sn_k, b.sn2_q = (_np(m, dev) for m in (a.proj_sn, a.sn_q, a.sn_k, a.sn2_q))
        b.attn_sn = _np(a.attn_sn, dev)
        return b
    blk = packed(mod, build)
    with torch.no_grad():
        x = _cl(x)
        Tq, B_, Wh, Ww, Cc = x.shape
        N1, rows = Wh * Ww, B_ * Wh * Ww
        M, n = Tq * rows, rows * Cc
        xs = torch.empty((M, Cc), dtype=torch.uint8, device=x.device)
        hip.neuron_fwd(x, xs, Tq, 1, n, 0, n, 0, n, blk.sn_proj)                                   # proj_sn (:670)
        q = torch.empty((M, Cc), dtype=torch.uint8, device=x.device)
        k = torch.empty((M, Cc), dtype=torch.uint8, device=x.device)
        hip.spike_gemm_sn(xs, blk.q.Wp, q, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_q, alpha=blk.q.alpha, beta=blk.q.beta)
        hip.spike_gemm_sn(xs, blk.k.Wp, k, Cc, Cc, Tq, rows, rows, 0, rows, blk.sn_k, alpha=blk.k.alpha, beta=blk.k.beta,
                          add=blk.pe.reshape(-1, Cc), add_prows=N1)                                 # + positional term (:678-680)
        e = torch.empty((M, Cc), dtype=torch.uint8, device=x.device)
        hip.qk_gate(q, k, e, Tq, rows, Cc, blk.sn2_q)                                               # :687-694
        out = torch.empty((M, Cc), dtype=torch.float32, device=x.device)
        hip.spike_gemm(e, blk.p.Wp, out, M, Cc, Cc, bias=blk.p.bias, alpha=blk.p.alpha, beta=blk.p.beta,
                       zg=(blk.nH, Tq, B_, N1))                                                      # head scramble + proj + BN (:709-714)
        from .engine import attention_score
        score = attention_score(e.view(Tq, rows, Cc), blk.attn_sn, blk.nH, Tq, B_, N1).view(Tq, B_, Wh, Ww, Cc)
        return out.view(B_, Tq * N1, Cc), score                                                     # the raw reshape of :715


def ms_patch_merging_forward(mod, x):
    """`MS_SpikingPatchMerging.forward` (reference :952-974): (B,D,H,W,C) -> (B,D,H/2,W/2,2C)."""
    from .engine import MSFlowEngine, _Lin, _np
    _eval_only(mod)
    eng, lin, sn = packed(mod, lambda dev: (MSFlowEngine.bare(dev), _Lin(mod.reduction, mod.norm.norm_layer, dev, 2), _np(mod.sn, dev)))
    with torch.no_grad():
        return eng.patch_merge(_cl(x), 0, packed=(lin, sn))


# ------------------------------------------------------------------------------------------------ SEW family
def ms_resblock_forward(mod, x):
    """`MS_ResBlock.forward` (reference Spiking_modules.py:906-933): x (T,B,C,H,W) membrane -> SN - conv - BN - SN - conv - BN + identity,
    same shape.  The shipped form only: BatchNorm behind both convolutions, the ADD shortcut, stride 1."""
    from .engine import MSFlowEngine, _ResBlock
    _eval_only(mod)
    if mod.norm is None or mod.connect_function != "ADD" or mod.conv1[0].stride != (1, 1):
        raise NotImplementedError("MS_ResBlock.forward at module level: spike_norm='BN', connect_function='ADD', stride 1 (the shipped form)")
    eng, rb = packed(mod, lambda dev: (MSFlowEngine.bare(dev), _ResBlock(mod, dev, 2)))
    with torch.no_grad():
        m = _cl(x).permute(1, 0, 3, 4, 2).contiguous()                 # (B, T, H, W, C): the engine's channel-last membrane
        return eng._resblock(m, rb).permute(1, 0, 4, 2, 3).contiguous()


def _sew(mod_block):
    from .engine_sew import SEWFlowEngine, _SewBlock
    return packed(mod_block, lambda dev: (SEWFlowEngine.bare(dev), _SewBlock(mod_block, dev, 2, "")))


def sew_block_forward(mod, x, mask_matrix=None, return_attention=False):
    """`Spiking_SwinTransformerBlock3D.forward` (reference :824-847): x (B,D,H,W,C) -> SSA(x) + x, then MLP(.) + (.).  The shift
    mask is derived from the shape like the layer does (:1070-1076); a passed `mask_matrix` is accepted for the signature."""
    _eval_only(mod)
    if return_attention:
        raise NotImplementedError("return_attention is not built")
    eng, blk = _sew(mod)
    with torch.no_grad():
        x = _cl(x)
        x = eng.attention(x, blk) + x
        return eng.mlp(x, blk) + x


def sew_mlp_forward(mod, x):
    """`Spiking_Mlp.forward` (reference :147-162): x (T,B,H,W,C) -> spikes of the same shape."""
    from .engine import _Lin, _np, bn_affine
    from .engine_sew import SEWFlowEngine
    _eval_only(mod)

    def build(dev):
        class P:
            pass
        b = P()
        b.name = ""
        b.w1 = hip.pack_dense_linear_weight(mod.fc1.weight.detach().float().to(dev))
        b.bn1, b.sn1 = bn_affine(mod.bn1.norm_layer, dev), _np(mod.sn1, dev)
        b.fc2, b.sn2 = _Lin(mod.fc2, mod.bn2.norm_layer, dev, 2), _np(mod.sn2, dev)
        return SEWFlowEngine.bare(dev), b
    eng, b = packed(mod, build)
    with torch.no_grad():
        y = eng.mlp(_cl(x).permute(1, 0, 2, 3, 4).contiguous(), b)
        return y.permute(1, 0, 2, 3, 4).contiguous()


def sew_patch_merging_forward(mod, x):
    """`SpikingPatchMerging.forward` (reference :914-934)."""
    from .engine import _np, bn_affine
    from .engine_sew import SEWFlowEngine
    _eval_only(mod)
    eng, pk = packed(mod, lambda dev: (SEWFlowEngine.bare(dev), (hip.pack_dense_linear_weight(mod.reduction.weight.detach().float().to(dev)),
                                                                 bn_affine(mod.norm.norm_layer, dev), _np(mod.sn, dev))))
    with torch.no_grad():
        return eng.patch_merge(_cl(x), 0, packed=pk)
