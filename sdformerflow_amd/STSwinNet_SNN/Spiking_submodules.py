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
