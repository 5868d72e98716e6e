"""CPU restatement of the neurons' BACKWARD (TEST INFRASTRUCTURE ONLY - the checker of csrc/neuron_bwd.hip).

What the reference differentiates (training path):
  * `neuron.LIFNode` / `IFNode` of spikingjelly 0.0.0.0.14 in multi-step mode on the torch backend, i.e. plain autograd
    through the per-step loop `h = v + (x - v)/tau ; s = surrogate(h - v_th) ; v = h - s.detach()*v_th` (built in
    Spiking_modules.py:40-66; `detach_reset` and `surrogate.ATan()` come from configs/*.yml).  The package is not in this
    image: the equations are its published ones (oracle/stubs/README.md, "parity unpinned" at that boundary) and the
    fixtures of tests/golden/neuron_grads.npz were produced by the reference's own `Spiking_neuron` factory on them.
  * `PSN.forward` (Spiking_submodules.py:207-211): `h = addmm(bias, weight, x.flatten(1)) ; s = surrogate(h)` - in-tree
    code, pinned by import.
  * `surrogate.ATan`: forward heaviside(x >= 0), backward `alpha / 2 / (1 + (pi/2 * alpha * x)^2) * grad`.

Written as explicit reverse-time recurrences in fp32 torch ops (no autograd), in the operation order autograd applies,
so that the HIP kernel can be compared bit for bit; tests/test_oracle_golden.py checks it against the fixtures.
"""
import math

import torch


def atan_grad(u, g, alpha=2.0):
    """reference surrogate.ATan backward; torch evaluates `a / 2 / y * g` as `(y.reciprocal() * (a/2)) * g`."""
    t = (math.pi / 2 * alpha) * u
    y = 1 + t * t
    return (y.reciprocal() * (alpha / 2)) * g


def lif_forward_h(x, tau, v_th, v_reset, kind="lif"):
    """Membrane before fire h_t (T, ...) and spikes, same arithmetic as oracle/csrc/neuron_ref.c."""
    soft = v_reset is None
    v = torch.zeros_like(x[0]) if soft else torch.full_like(x[0], v_reset)
    hs, ss = [], []
    for t in range(x.shape[0]):
        if kind == "if":
            h = v + x[t]
        elif soft or v_reset == 0.0:
            h = v + (x[t] - v) / tau
        else:
            h = v + (x[t] - (v - v_reset)) / tau
        s = (h - v_th >= 0).to(x.dtype)
        v = h - s * v_th if soft else (1.0 - s) * h + s * v_reset
        hs.append(h)
        ss.append(s)
    return torch.stack(hs), torch.stack(ss)


def lif_backward(x, grad_spike, tau=2.0, v_th=1.0, v_reset=None, detach_reset=True, alpha=2.0, kind="lif"):
    """dL/dx (T, ...) of the multi-step LIF / IF given dL/dspike: BPTT, final membrane unused."""
    x, grad_spike = x.float(), grad_spike.float()
    h, s = lif_forward_h(x, tau, v_th, v_reset, kind)
    soft = v_reset is None
    gv = torch.zeros_like(x[0])
    gx = torch.empty_like(x)
    for t in range(x.shape[0] - 1, -1, -1):
        u = h[t] - v_th
        gsp = grad_spike[t]
        if soft:
            if not detach_reset:
                gsp = gsp + (-(gv * v_th))
            gh = gv + atan_grad(u, gsp, alpha)
        else:
            if not detach_reset:
                gsp = gsp + (gv * v_reset + (-(gv * h[t])))
            gh = gv * (1.0 - s[t]) + atan_grad(u, gsp, alpha)
        if kind == "if":
            gx[t], gv = gh, gh
        else:
            q = gh / tau
            gx[t], gv = q, gh - q
    return gx


def psn_backward(x, W, b, grad_spike, alpha=2.0):
    """(dL/dx, dL/dW, dL/db) of PSN.forward; h through torch.addmm as the reference computes it."""
    T = x.shape[0]
    xf = x.float().reshape(T, -1)
    h = torch.addmm(b.float().reshape(T, 1), W.float(), xf)
    gh = atan_grad(h, grad_spike.float().reshape(T, -1), alpha)
    return (W.float().t() @ gh).view(x.shape), gh @ xf.t(), gh.sum(1, keepdim=True)
