"""Neuron parameter holders with the traversal surface the reference's harness relies on
(`functional.reset_net / set_step_mode / set_backend`, reference eval_DSEC_flow_SNN.py:101-119,155):
`reset()`, `step_mode`, `backend`, `supported_backends`, `store_v_seq`, `v`.

The arithmetic lives in csrc/neuron.hip; these classes only carry hyper-parameters / weights and call
the C ABI when used stand-alone on a GPU tensor.  Mirrors reference models/STSwinNet_SNN/Spiking_submodules.py
(PSN :183-211) and the spikingjelly LIFNode/IFNode the reference instantiates (Spiking_modules.py:40-66).
"""
import math

import torch
import torch.nn as nn

from .. import hip


class _NodeBase(nn.Module):
    supported_backends = ("torch", "cupy", "hip")

    def __init__(self):
        super().__init__()
        self.step_mode = "m"
        self.backend = "hip"
        self.store_v_seq = False
        self.v = 0.0

    def reset(self):
        self.v = 0.0 if getattr(self, "v_reset", None) is None else self.v_reset


class LIFNode(_NodeBase):
    kind = "lif"

    def __init__(self, tau=2.0, v_threshold=1.0, v_reset=0.0, surrogate_function=None, detach_reset=False):
        super().__init__()
        self.tau, self.v_threshold, self.v_reset = float(tau), float(v_threshold), v_reset
        self.surrogate_function, self.detach_reset = surrogate_function, detach_reset
        self.reset()

    def params(self, device=None):
        return hip.NeuronParams(self.kind, self.tau, self.v_threshold, self.v_reset)

    def forward(self, x_seq):
        if torch.is_grad_enabled() and x_seq.requires_grad:        # training path: HIP forward + HIP BPTT backward
            from ..autograd import LIFFunction
            return LIFFunction.apply(x_seq, self.tau, self.v_threshold, self.v_reset, self.detach_reset,
                                     getattr(self.surrogate_function, "alpha", 2.0), self.kind)
        out, v = hip.lif_fwd(x_seq, self.tau, self.v_threshold, self.v_reset, torch.float32, return_v=True)
        self.v = v
        return out

    def extra_repr(self):
        return f"tau={self.tau}, v_threshold={self.v_threshold}, v_reset={self.v_reset}, backend=hip"


class IFNode(LIFNode):
    kind = "if"

    def __init__(self, v_threshold=1.0, v_reset=0.0, surrogate_function=None, detach_reset=False):
        super().__init__(2.0, v_threshold, v_reset, surrogate_function, detach_reset)

    def forward(self, x_seq):
        raise hip.SdfError("IFNode is only reachable through preds_out, which the reference never calls "
                           "(Spiking_STSwinNet.py:154-156)")


class SLTTLIFNode(LIFNode):
    """reference Spiking_submodules.py:11-90.  "The forward propagation is the same as the Leaky Integrate-and-Fire neuron's"
    (:17-19): inference runs on the LIF kernels.  What differs is the gradient (the membrane is detached between steps, :41),
    which the training path here does not build."""

    def forward(self, x_seq):
        if torch.is_grad_enabled() and x_seq.requires_grad:
            raise NotImplementedError("SLTTLIFNode: the online (detached-membrane) gradient is not built; inference only")
        return super().forward(x_seq)


class ParametricLIFNode(LIFNode):
    """spikingjelly `neuron.ParametricLIFNode` as the reference instantiates it (Spiking_modules.py:75-82): a LIF whose charge is
    v += (x - (v - v_reset)) * sigmoid(w), w learnable, init w = -log(init_tau - 1).  The kernels take the multiplier k = sigmoid(w)
    through the `tau` field of the C ABI (0 < tau < 1 selects the multiplicative charge, csrc/common.h `sdf_inv_tau`), so every fused
    kernel of the engine runs this neuron exactly; `tau` below is that k."""

    def __init__(self, init_tau=2.0, v_threshold=1.0, v_reset=0.0, surrogate_function=None, detach_reset=False):
        assert isinstance(init_tau, float) and init_tau > 1.0
        _NodeBase.__init__(self)
        self.v_threshold, self.v_reset = float(v_threshold), v_reset
        self.surrogate_function, self.detach_reset = surrogate_function, detach_reset
        self.reset()
        self.w = nn.Parameter(torch.as_tensor(-math.log(init_tau - 1.0), dtype=torch.float32))

    @property
    def tau(self):
        k = float(torch.sigmoid(self.w.detach().float()))
        if not 0.0 < k < 1.0:
            raise hip.SdfError(f"ParametricLIFNode: sigmoid(w) = {k} is not inside (0, 1)")
        return k

    def forward(self, x_seq):
        if torch.is_grad_enabled() and (x_seq.requires_grad or self.training):
            raise NotImplementedError("ParametricLIFNode: the gradient (to x and to w) is not built; inference only")
        return super().forward(x_seq)

    def extra_repr(self):
        return f"k=sigmoid(w)={self.tau}, v_threshold={self.v_threshold}, v_reset={self.v_reset}, backend=hip"


class GatedLIFNode(nn.Module):
    """reference Spiking_submodules.py:94-181 (GLIF, layer-wise gates as `Spiking_neuron` builds it: `inplane=None`,
    Spiking_modules.py:84-92).  No HIP kernel: the recurrence below is torch element-wise ops on the tensor's own device (what
    SURVEY.md section 2 row 4 asks: API-complete, torch fallback), so the module works stand-alone; the fused engines refuse a
    model built with it (no shipped configuration uses it).  Same operation order as the reference's `multi_step_forward`."""
    kind = "glif"
    supported_backends = ("torch",)

    def __init__(self, T, inplane=None, init_linear_decay=None, init_v_subreset=None, init_tau=0.25, init_v_threshold=0.5,
                 init_conduct=0.5, surrogate_function=None, step_mode="m", backend="torch"):
        super().__init__()
        assert isinstance(init_tau, float) and init_tau < 1.0 and isinstance(T, int) and step_mode == "m"
        if inplane is not None:
            raise NotImplementedError("channel-wise GLIF gates: the reference only builds the layer-wise form (Spiking_modules.py:84-92)")
        self.T, self.surrogate_function, self.step_mode, self.backend = T, surrogate_function, step_mode, backend
        logit = lambda p: -math.log(1.0 / p - 1.0)
        self.alpha, self.beta, self.gamma = (nn.Parameter(0.2 * (torch.rand(()) - 0.5)) for _ in range(3))
        self.tau = nn.Parameter(torch.tensor(logit(init_tau)))
        self.v_threshold = nn.Parameter(torch.tensor(logit(init_v_threshold)))
        self.linear_decay = nn.Parameter(torch.tensor(logit(init_v_threshold / (T * 2) if init_linear_decay is None else init_linear_decay)))
        self.v_subreset = nn.Parameter(torch.tensor(logit(init_v_threshold if init_v_subreset is None else init_v_subreset)))
        self.conduct = nn.Parameter(logit(init_conduct) * torch.ones(T))
        self.v = self.u = 0.0

    def reset(self):
        self.v = self.u = 0.0

    def forward(self, x_seq):
        if torch.is_grad_enabled() and (x_seq.requires_grad or self.training):
            raise NotImplementedError("GatedLIFNode: inference only (the surrogate gradient is not built)")
        if x_seq.shape[0] != self.T:
            raise hip.SdfError(f"GatedLIFNode(T={self.T}) got {x_seq.shape[0]} steps")
        if not x_seq.is_cuda:
            raise hip.SdfError("GatedLIFNode runs on the GPU tensor's device only (no CPU path in the product; the CPU restatement "
                               "is oracle.sdformer_oracle.glif_multistep)")
        with torch.no_grad():
            al, be, ga = self.alpha.sigmoid(), self.beta.sigmoid(), self.gamma.sigmoid()
            leak = 1 - al * (1 - self.tau.sigmoid())
            v, spike, out = self.v, torch.zeros_like(x_seq[0]), []
            for t in range(self.T):
                inp = x_seq[t] * (1 - be * (1 - self.conduct[t].sigmoid()))                     # neuronal_charge :155-159
                u = (leak * v - (1 - al) * self.linear_decay.sigmoid()) + inp
                u = u - leak * v * ga * spike - (1 - ga) * self.v_subreset.sigmoid() * spike     # neuronal_reset  :163-165
                spike = (u - self.v_threshold.sigmoid() >= 0).to(x_seq.dtype)                    # neuronal_fire   :169
                v = u
                out.append(spike)
            self.v = self.u = v
        return torch.stack(out)

    def extra_repr(self):
        return f"T={self.T}, backend=torch"


class PSN(nn.Module):
    """Parallel Spiking Neuron: H = bias + weight @ X over time, S = (H >= 0)."""
    kind = "psn"
    supported_backends = ("torch", "hip")

    def __init__(self, T, surrogate_function=None):
        super().__init__()
        self.T = T
        self.surrogate_function = surrogate_function
        self.step_mode, self.backend = "m", "hip"
        self.weight = nn.Parameter(torch.zeros(T, T))
        self.bias = nn.Parameter(torch.zeros(T, 1))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        nn.init.constant_(self.bias, -1.0)

    def params(self, device=None):
        return hip.NeuronParams("psn", psn_w=self.weight.detach().contiguous(),
                                psn_b=self.bias.detach().reshape(-1).contiguous())

    def forward(self, x_seq):
        if torch.is_grad_enabled() and (x_seq.requires_grad or self.weight.requires_grad):
            from ..autograd import PSNFunction
            return PSNFunction.apply(x_seq, self.weight, self.bias, getattr(self.surrogate_function, "alpha", 2.0))
        return hip.psn_fwd(x_seq, self.weight.detach(), self.bias.detach(), torch.float32)

    def extra_repr(self):
        return f"T={self.T}, backend=hip"
