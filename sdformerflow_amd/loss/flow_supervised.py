"""Evaluation metrics of the reference (loss/flow_supervised.py): AEE (+ PE1/2/3, outliers) :108-149, AAE :152-175."""
import math

import torch


class AEE(torch.nn.Module):
    """Average end-point error over valid pixels; call the instance to get (AEE[B], PE1, PE2, PE3, outliers)."""

    def __init__(self, pred, label, mask, flow_scaling=128):
        super().__init__()
        self.flow, self.label, self.mask, self.flow_scaling = pred, label, mask, flow_scaling

    def forward(self):
        flow = self.flow * self.flow_scaling
        B = flow.shape[0]
        mask = self.mask.reshape(B, -1)
        err = (flow - self.label).pow(2).sum(1).sqrt().view(B, -1) * mask
        mag = flow.pow(2).sum(1).sqrt().view(B, -1) * mask
        n = mask.sum(dim=1)
        aee = err.sum(dim=1) / (n + 1e-9)
        outliers = ((err > 3.0) * (err > 0.05 * mag)).sum() / (n + 1e-9)
        pe = [(err > th).sum() / (n + 1e-9) for th in (1.0, 2.0, 3.0)]
        return aee, pe[0], pe[1], pe[2], outliers


class AAE(torch.nn.Module):
    """Average angular error in degrees; returns a 1-tuple like the reference (:175)."""

    def __init__(self, pred, label, mask, flow_scaling=128):
        super().__init__()
        self.flow, self.label, self.mask, self.flow_scaling = pred, label, mask, flow_scaling

    def forward(self):
        flow = self.flow * self.flow_scaling
        fm = flow.pow(2).sum(1).sqrt() * self.mask
        gm = self.label.pow(2).sum(1).sqrt() * self.mask
        dot = flow[:, 0] * self.label[:, 0] + flow[:, 1] * self.label[:, 1]
        cos = torch.clamp((dot + 1e-7) / (fm * gm + 1e-7), min=-1.0 + 1e-7, max=1.0 - 1e-7)
        return (torch.sum(torch.acos(cos) * self.mask) / torch.sum(self.mask) * 180 / math.pi,)
