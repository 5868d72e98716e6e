"""GPU parity of the one-launch first half of the spiking QK window attention (csrc/qk_front.hip) through the C ABI
`sdf_qk_attn_fwd`: slice neuron + q | k projection with BN, positional term and neurons + token gate (reference
Spiking_swin_transformer3D.py:670-694, 789-804) in one kernel, against the SAME entry point's four-launch form (flag
SDF_QK_FOUR_LAUNCHES), which tests/test_hip_kernels.py ties to the individually tested entry points and tests/test_engine_gpu.py /
test_replay_gpu.py to the oracle.  The kernel issues its matrix products in the k order and plane order of the GEMM kernel it
replaces, so q, k, the gated spikes E and the block's output are BIT-EQUAL - for the stacked (lif) and the separate (psn)
projection form, every stage width of the shipped model, a window of 8 x 8 tokens, padded and shifted maps."""
import pytest
import torch

from sdformerflow_amd import hip
from sdformerflow_amd.engine import _Block
from sdformerflow_amd.STSwinNet_SNN import Spiking_swin_transformer3D as SW
from sdformerflow_amd.synthetic import synth_state_dict
from sdformerflow_amd.synthetic import synth_uniform as rnd

import os

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _front_on_every_width(monkeypatch):
    """Wide stages too (the dispatcher keeps three launches from C = 288 on); scoped to this file's tests - the library reads the
    variable per call, so the rest of the suite runs the SHIPPED dispatch."""
    monkeypatch.setenv("SDF_QK_FRONT_ANY", "1")
    monkeypatch.setenv("SDF_WIDE", "0")             # (the wide-stage kernels would take C >= 256 away from the kernel under test)


def _block(kind, Cc, nH, H, W, window, shift):
    kw = {"num_steps": 4, "v_reset": None, "v_th": 0.1, "neuron_type": kind, "surrogate_fun": "surrogate.ATan()", "tau": 2.0,
          "detach_reset": True, "spike_norm": "BN"}
    m = SW.MS_Spiking_SwinTransformerBlock3D(Cc, (H, W), nH, window_size=window, shift_size=shift, norm_layer="BN", **kw)
    m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}))
    return _Block(m.to(DEV).eval(), torch.device(DEV), 2)


def _run(blk, x0, rowmap, B_, Tq, N1, nH, four):
    x, keep = x0.clone(), []
    kw = dict(qk=blk.qk) if blk.qk is not None else dict(q_lin=blk.q, k_lin=blk.k, pe=blk.pe)
    hip.qk_attn(x, rowmap, B_, Tq, N1, nH, blk.p, blk.sn_proj, blk.sn_q, blk.sn_k, blk.sn2_q, keep_ws=keep, four_launches=four, **kw)
    torch.cuda.synchronize()
    Cc = x.shape[-1]
    M = Tq * B_ * N1
    ws = keep[0]
    e = ws[:M * Cc].clone()
    qk = ws[(M * Cc + 255) // 256 * 256:][:M * 2 * Cc].clone()
    return x, e, qk


@pytest.mark.parametrize("kind", ["lif", "psn"])
@pytest.mark.parametrize("B,D,H,W,Cc,nH,window,shift", [
    (2, 4, 18, 21, 96, 3, (2, 9, 9), (1, 4, 4)),          # padded width, shifted
    (1, 10, 36, 48, 192, 6, (2, 9, 9), (0, 0, 0)),
    (1, 10, 18, 24, 384, 12, (2, 9, 9), (1, 4, 4)),
    (1, 10, 9, 12, 768, 24, (2, 9, 9), (0, 0, 0)),
    (1, 4, 16, 24, 96, 3, (2, 8, 8), (1, 4, 4)),          # 64-token slices: two token blocks
])
def test_one_launch_first_half_is_bit_equal_to_the_three_launches(kind, B, D, H, W, Cc, nH, window, shift):
    blk = _block(kind, Cc, nH, H, W, window, shift)
    assert (blk.qk is not None) == (kind == "lif")
    x0 = rnd((B, D, H, W, Cc), 31 + Cc, -0.5, 1.0).to(DEV)
    rowmap, B_ = hip.window_slice_map(B, D, H, W, window, shift, DEV)
    Tq, N1 = window[0], window[1] * window[2]
    xa, ea, qka = _run(blk, x0, rowmap, B_, Tq, N1, nH, four=True)
    xb, eb, qkb = _run(blk, x0, rowmap, B_, Tq, N1, nH, four=False)
    assert torch.equal(qka, qkb), f"q | k spikes differ in {(qka != qkb).sum().item()} of {qka.numel()} bytes"
    assert torch.equal(ea, eb), f"gated spikes differ in {(ea != eb).sum().item()} of {ea.numel()} bytes"
    assert torch.equal(xa, xb) and not torch.equal(xa, x0)
    assert 0.01 < ea.float().mean().item() < 0.9 and 0.02 < qka.float().mean().item() < 0.98
    # without the tape the one-launch form writes E only, and the block's output is the same
    xc = x0.clone()
    kw = dict(qk=blk.qk) if blk.qk is not None else dict(q_lin=blk.q, k_lin=blk.k, pe=blk.pe)
    hip.qk_attn(xc, rowmap, B_, Tq, N1, nH, blk.p, blk.sn_proj, blk.sn_q, blk.sn_k, blk.sn2_q, **kw)
    torch.cuda.synchronize()
    assert torch.equal(xc, xb)
