"""Stub of spikingjelly.activation_based.surrogate (own code)."""
import math
import torch
import torch.nn as nn


def heaviside(x):
    return (x >= 0).to(x)


class SurrogateFunctionBase(nn.Module):
    def __init__(self, alpha, spiking=True):
        super().__init__()
        self.spiking = spiking
        self.alpha = alpha

    def forward(self, x):
        if self.spiking:
            return self.spiking_function(x, self.alpha)
        return self.primitive_function(x, self.alpha)


class _ATanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return heaviside(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        a = ctx.alpha
        return a / 2 / (1 + (math.pi / 2 * a * x).pow(2)) * g, None


class ATan(SurrogateFunctionBase):
    def __init__(self, alpha=2.0, spiking=True):
        super().__init__(alpha, spiking)

    @staticmethod
    def spiking_function(x, alpha):
        return _ATanFn.apply(x, alpha)

    @staticmethod
    def primitive_function(x, alpha):
        return (math.pi / 2 * alpha * x).atan() / math.pi + 0.5


class _SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, alpha):
        ctx.save_for_backward(x)
        ctx.alpha = alpha
        return heaviside(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        s = torch.sigmoid(ctx.alpha * x)
        return g * s * (1 - s) * ctx.alpha, None


class Sigmoid(SurrogateFunctionBase):
    def __init__(self, alpha=4.0, spiking=True):
        super().__init__(alpha, spiking)

    @staticmethod
    def spiking_function(x, alpha):
        return _SigmoidFn.apply(x, alpha)

    @staticmethod
    def primitive_function(x, alpha):
        return torch.sigmoid(alpha * x)
