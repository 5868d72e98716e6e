"""Stub of spikingjelly.activation_based.functional (own code)."""
import torch


def reset_net(net):
    for m in net.modules():
        if hasattr(m, "reset"):
            m.reset()


def set_step_mode(net, step_mode):
    from . import base
    for m in net.modules():
        if isinstance(m, base.StepModule) or hasattr(m, "step_mode"):
            try:
                m.step_mode = step_mode
            except ValueError:
                pass


def set_backend(net, backend, instance=None):
    for m in net.modules():
        if instance is not None and not isinstance(m, instance):
            continue
        if hasattr(m, "backend") and backend in getattr(m, "supported_backends", ()):
            m.backend = backend


def seq_to_ann_forward(x_seq, stateless):
    y = x_seq.flatten(0, 1)
    if isinstance(stateless, (list, tuple)):
        for m in stateless:
            y = m(y)
    else:
        y = stateless(y)
    return y.view([x_seq.shape[0], x_seq.shape[1]] + list(y.shape[1:]))
