"""ctypes front-end of oracle/csrc/neuron_ref.c (TEST INFRASTRUCTURE ONLY).

Exact fp32 semantics (true `fmaf`, separately rounded LIF ops) that torch on CPU cannot express;
used by the GPU parity tests as the bit-exact checker of the neuron kernels.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libneuron_ref.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _prep(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def neuron_ref(x, kind="lif", tau=2.0, v_th=1.0, v_reset=None, psn_w=None, psn_b=None, alpha=None, beta=None,
               inner=1, add=None, add_period=0, return_aux=False):
    """x (T, ...) fp32 on CPU -> spikes (same shape, fp32).  `alpha/beta` (C,) apply fmaf per channel
    c = (n // inner) % C over the flattened per-step index n; `add` is (T, add_period)."""
    x = _prep(x)
    T = x.shape[0]
    N = x[0].numel()
    s = torch.empty_like(x)
    alpha, beta, add = _prep(alpha), _prep(beta), _prep(add)
    C = alpha.numel() if alpha is not None else 1
    add_st = add_period if add is not None else 0
    L = _lib()
    if kind == "psn":
        h = torch.empty_like(x)
        L.ref_psn(_p(x), _p(_prep(psn_w)), _p(_prep(psn_b).reshape(-1)), _p(s), _p(h), ctypes.c_int(T),
                  ctypes.c_int64(N), _p(alpha), _p(beta), ctypes.c_int(C), ctypes.c_int64(inner), _p(add),
                  ctypes.c_int64(add_st), ctypes.c_int64(max(add_period, 1)))
        return (s, h) if return_aux else s
    v = torch.empty(N, dtype=torch.float32)
    soft = v_reset is None
    L.ref_lif(_p(x), _p(s), _p(v), ctypes.c_int(T), ctypes.c_int64(N), ctypes.c_float(tau), ctypes.c_float(v_th),
              ctypes.c_int(1 if soft else 0), ctypes.c_float(0.0 if soft else v_reset), _p(alpha), _p(beta),
              ctypes.c_int(C), ctypes.c_int64(inner), _p(add), ctypes.c_int64(add_st),
              ctypes.c_int64(max(add_period, 1)), ctypes.c_int(1 if kind == "if" else 0))
    return (s, v.view(x.shape[1:])) if return_aux else s
