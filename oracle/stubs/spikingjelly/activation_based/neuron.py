"""Stub of spikingjelly.activation_based.neuron (own code).

Equations follow SpikingJelly 0.0.0.0.14's public documentation; the package source is not
available in this image, so this boundary is "parity unpinned" (see oracle/stubs/README.md).
"""
import math
import torch
import torch.nn as nn
from . import base, surrogate  # noqa: F401  (reference imports `surrogate, base` from here)


class BaseNode(base.MemoryModule):
    def __init__(self, v_threshold=1.0, v_reset=0.0, surrogate_function=None, detach_reset=False,
                 step_mode="s", backend="torch", store_v_seq=False):
        super().__init__()
        if surrogate_function is None:
            surrogate_function = surrogate.Sigmoid()
        self.register_memory("v", 0.0 if v_reset is None else v_reset)
        self.v_threshold = v_threshold
        self.v_reset = v_reset
        self.detach_reset = detach_reset
        self.surrogate_function = surrogate_function
        self.step_mode = step_mode
        self.backend = backend
        self.store_v_seq = store_v_seq

    @property
    def supported_backends(self):
        return ("torch", "cupy")

    def neuronal_charge(self, x):
        raise NotImplementedError

    def neuronal_fire(self):
        return self.surrogate_function(self.v - self.v_threshold)

    def neuronal_reset(self, spike):
        s = spike.detach() if self.detach_reset else spike
        if self.v_reset is None:
            self.v = self.v - s * self.v_threshold
        else:
            self.v = (1.0 - s) * self.v + s * self.v_reset

    def v_float_to_tensor(self, x):
        if isinstance(self.v, float):
            self.v = torch.full_like(x.data, self.v)

    def single_step_forward(self, x):
        self.v_float_to_tensor(x)
        self.neuronal_charge(x)
        spike = self.neuronal_fire()
        self.neuronal_reset(spike)
        return spike

    def multi_step_forward(self, x_seq):
        ys, vs = [], []
        for t in range(x_seq.shape[0]):
            ys.append(self.single_step_forward(x_seq[t]))
            if self.store_v_seq:
                vs.append(self.v)
        if self.store_v_seq:
            self.v_seq = torch.stack(vs)
        return torch.stack(ys)


class IFNode(BaseNode):
    def neuronal_charge(self, x):
        self.v = self.v + x


class LIFNode(BaseNode):
    def __init__(self, tau=2.0, decay_input=True, v_threshold=1.0, v_reset=0.0, surrogate_function=None,
                 detach_reset=False, step_mode="s", backend="torch", store_v_seq=False):
        assert isinstance(tau, float) and tau > 1.0
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset, step_mode, backend, store_v_seq)
        self.tau = tau
        self.decay_input = decay_input

    @staticmethod
    def neuronal_charge_decay_input_reset0(x, v, tau):
        return v + (x - v) / tau

    @staticmethod
    def neuronal_charge_decay_input(x, v, v_reset, tau):
        return v + (x - (v - v_reset)) / tau

    @staticmethod
    def neuronal_charge_no_decay_input_reset0(x, v, tau):
        return v * (1.0 - 1.0 / tau) + x

    @staticmethod
    def neuronal_charge_no_decay_input(x, v, v_reset, tau):
        return v - (v - v_reset) / tau + x

    def neuronal_charge(self, x):
        r0 = self.v_reset is None or self.v_reset == 0.0
        if self.decay_input:
            self.v = (self.neuronal_charge_decay_input_reset0(x, self.v, self.tau) if r0
                      else self.neuronal_charge_decay_input(x, self.v, self.v_reset, self.tau))
        else:
            self.v = (self.neuronal_charge_no_decay_input_reset0(x, self.v, self.tau) if r0
                      else self.neuronal_charge_no_decay_input(x, self.v, self.v_reset, self.tau))

    # eval-mode helpers SLTTLIFNode expects (reference Spiking_submodules.py:72-90)
    @staticmethod
    def jit_eval_single_step_forward_soft_reset_decay_input(x, v, v_th, tau):
        v = v + (x - v) / tau
        s = (v >= v_th).to(x)
        return s, v - s * v_th

    @staticmethod
    def jit_eval_single_step_forward_soft_reset_no_decay_input(x, v, v_th, tau):
        v = v * (1.0 - 1.0 / tau) + x
        s = (v >= v_th).to(x)
        return s, v - s * v_th

    @staticmethod
    def jit_eval_single_step_forward_hard_reset_decay_input(x, v, v_th, v_reset, tau):
        v = v + (x - (v - v_reset)) / tau
        s = (v >= v_th).to(x)
        return s, v_reset * s + (1.0 - s) * v

    @staticmethod
    def jit_eval_single_step_forward_hard_reset_no_decay_input(x, v, v_th, v_reset, tau):
        v = v - (v - v_reset) / tau + x
        s = (v >= v_th).to(x)
        return s, v_reset * s + (1.0 - s) * v


class ParametricLIFNode(BaseNode):
    def __init__(self, init_tau=2.0, decay_input=True, v_threshold=1.0, v_reset=0.0, surrogate_function=None,
                 detach_reset=False, step_mode="s", backend="torch", store_v_seq=False):
        assert isinstance(init_tau, float) and init_tau > 1.0
        super().__init__(v_threshold, v_reset, surrogate_function, detach_reset, step_mode, backend, store_v_seq)
        self.decay_input = decay_input
        self.w = nn.Parameter(torch.as_tensor(-math.log(init_tau - 1.0)))

    def neuronal_charge(self, x):
        k = self.w.sigmoid()
        r0 = self.v_reset is None or self.v_reset == 0.0
        if self.decay_input:
            self.v = self.v + (x - self.v) * k if r0 else self.v + (x - (self.v - self.v_reset)) * k
        else:
            self.v = self.v * (1.0 - k) + x if r0 else self.v - (self.v - self.v_reset) * k + x
