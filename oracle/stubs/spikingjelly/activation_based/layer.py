"""Stub of spikingjelly.activation_based.layer (own code)."""
import torch
import torch.nn as nn
from . import base, functional


class _SeqMixin(base.StepModule):
    def _fwd(self, cls, x):
        if self.step_mode == "s":
            return cls.forward(self, x)
        if x.dim() != 5:
            raise ValueError(f"expected (T,B,C,H,W), got {tuple(x.shape)}")
        y = cls.forward(self, x.flatten(0, 1))
        return y.view([x.shape[0], x.shape[1]] + list(y.shape[1:]))


class Conv2d(nn.Conv2d, _SeqMixin):
    def __init__(self, *a, step_mode="s", **k):
        super().__init__(*a, **k)
        self.step_mode = step_mode

    def forward(self, x):
        return self._fwd(nn.Conv2d, x)


class ConvTranspose2d(nn.ConvTranspose2d, _SeqMixin):
    def __init__(self, *a, step_mode="s", **k):
        super().__init__(*a, **k)
        self.step_mode = step_mode

    def forward(self, x):
        return self._fwd(nn.ConvTranspose2d, x)


class BatchNorm2d(nn.BatchNorm2d, _SeqMixin):
    def __init__(self, *a, step_mode="s", **k):
        super().__init__(*a, **k)
        self.step_mode = step_mode

    def forward(self, x):
        return self._fwd(nn.BatchNorm2d, x)


class GroupNorm(nn.GroupNorm, _SeqMixin):
    def __init__(self, *a, step_mode="s", **k):
        super().__init__(*a, **k)
        self.step_mode = step_mode

    def forward(self, x):
        return self._fwd(nn.GroupNorm, x)


class Linear(nn.Linear, base.StepModule):
    def __init__(self, in_features, out_features, bias=True, step_mode="s"):
        super().__init__(in_features, out_features, bias)
        self.step_mode = step_mode


class Dropout(base.MemoryModule):
    def __init__(self, p=0.5, step_mode="s"):
        super().__init__()
        assert 0 <= p < 1
        self.step_mode = step_mode
        self.register_memory("mask", None)
        self.p = p

    def _make(self, x):
        self.mask = torch.nn.functional.dropout(torch.ones_like(x.data), self.p, training=True)

    def single_step_forward(self, x):
        if self.training:
            if self.mask is None:
                self._make(x)
            return x * self.mask
        return x

    def multi_step_forward(self, x_seq):
        if self.training:
            if self.mask is None:
                self._make(x_seq[0])
            return x_seq * self.mask
        return x_seq


class ThresholdDependentBatchNorm2d(BatchNorm2d):
    def __init__(self, alpha, v_th, *a, **k):
        super().__init__(*a, **k)
        self.alpha, self.v_th = alpha, v_th
        nn.init.constant_(self.weight, alpha * v_th)
