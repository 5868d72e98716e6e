"""ctypes binding of libsdformerflow_hip.so (the C ABI declared in include/sdformerflow_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every compute call goes through the
C ABI with raw pointers.  There is NO CPU fallback: if the library is missing or a tensor is not on
the GPU the call raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDF_HIP_LIB selects a DIAGNOSTIC build (tools/stamp_pp.sh, tools/gemm_ablate.sh build theirs beside the product library,
# never over it); unset = the product library
LIB_PATH = os.environ.get("SDF_HIP_LIB") or os.path.join(_HERE, "csrc", "libsdformerflow_hip.so")

SDF_F32, SDF_U8 = 0, 1
SDF_LIF, SDF_PSN, SDF_IF = 0, 1, 2
KIND = {"lif": SDF_LIF, "psn": SDF_PSN, "if": SDF_IF}

VOID_EXPORTS = ("sdf_switches_reload", "sdf_launch_log")       # the entry points that return nothing
EXPORTS = ("sdf_version", "sdf_switches_reload", "sdf_launch_log", "sdf_launch_log_read", "sdf_lif_fwd", "sdf_psn_fwd", "sdf_neuron_fwd", "sdf_spike_gemm_fwd",
           "sdf_split_weight_bf16", "sdf_split_weight_f16x2", "sdf_qk_gate_fwd", "sdf_qk_gate_strided_fwd", "sdf_affine_resid_fwd", "sdf_win_attn_fwd", "sdf_ann_attn_block_fwd", "sdf_ann_attn_block_supported", "sdf_ann_mlp_block_fwd", "sdf_ann_mlp_block_supported", "sdf_spike_conv2d_fwd", "sdf_spike_deconv3x3s2_fwd", "sdf_head_conv_sn_fwd",
           "sdf_flow_out_fwd", "sdf_deconv_col2im_fwd", "sdf_lif_bwd", "sdf_psn_bwd", "sdf_psn_bwd_workspace_bytes",
           "sdf_window_slice_map", "sdf_window_zsrc_map", "sdf_qk_attn_fwd", "sdf_qk_attn_is_wide", "sdf_ms_mlp_is_wide", "sdf_ms_patch_merge_fwd", "sdf_tile_weight_i8x3", "sdf_spike_conv2d_multi_fwd", "sdf_qk_attn_workspace_bytes", "sdf_spike_gemm_bn_fwd",
           "sdf_ms_mlp_fwd", "sdf_ms_mlp_workspace_bytes", "sdf_pred_head_fwd", "sdf_pointwise_conv_f32_fwd", "sdf_neuron_multi_fwd", "sdf_qk_gate_f32_fwd", "sdf_qk_gate_bwd", "sdf_qk_gate_bwd_workspace_bytes",
           "sdf_rows_gather_fwd", "sdf_rows_scatter_fwd",
           "sdf_split_weight_i8x3", "sdf_bn_train_fwd", "sdf_bn_train_bwd", "sdf_bn_train_workspace_bytes", "sdf_bn_train_nchw_fwd", "sdf_bn_train_nchw_bwd",
           "sdf_dense_conv3x3_fwd", "sdf_pack_planes", "sdf_unpack_planes", "sdf_pack_planes_up2", "sdf_pack_planes_zero_up2", "sdf_dense_linear_fwd", "sdf_layer_norm_fwd",
           "sdf_linear_dw_fwd", "sdf_linear_dw_splits", "sdf_ringed_rows_fwd", "sdf_linear_train_fwd", "sdf_unring_rows_fwd")


class SdfError(RuntimeError):
    """`rc` = the C ABI's return code when the error came from an entry point (negative: SDF_E_* argument refusal before any launch,
    positive: hipError_t), else None."""

    def __init__(self, msg, rc=None):
        super().__init__(msg)
        self.rc = rc


class ReplicaGeometryError(SdfError):
    """A replica launch sequence cannot serve this geometry (the window count per sample is not a multiple of the window depth at some
    stage): the caller runs the samples one by one - same results."""


# The diagnostic SDF_* switches are read ONCE - at import here, at the first call that asks for one in the
# library (csrc/switches.hip) - not per call.  A harness that changes the environment afterwards says so: reload_switches().
_SW = {k: v for k, v in os.environ.items() if k.startswith("SDF_")}


def sw(name, default=None):
    """Value of the switch SDF_<...> as of import / the last reload_switches() (a drop-in for os.environ.get)."""
    return _SW.get(name, default)


def reload_switches():
    """Re-read every SDF_* switch from the environment, on both sides of the C ABI (tests/conftest.py calls it around setenv)."""
    _SW.clear()
    _SW.update({k: v for k, v in os.environ.items() if k.startswith("SDF_")})
    if _lib is not None:
        _lib.sdf_switches_reload()


class scoped_switches:
    """`with hip.scoped_switches(SDF_CONV_WRES="0"):` - set (value None: unset) diagnostic switches for a block of calls and restore
    them afterwards, refreshing the tables both times (A/B legs of tests and tools)."""

    def __init__(self, **kw):
        self._kw = kw

    def __enter__(self):
        self._old = {k: os.environ.get(k) for k in self._kw}
        for k, v in self._kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        reload_switches()
        return self

    def __exit__(self, *exc):
        for k, v in self._old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        reload_switches()


class NeuronDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("v_last", C.c_void_p),
                ("T", C.c_int32), ("out_dtype", C.c_int32),
                ("nb", C.c_int64), ("ni", C.c_int64),
                ("x_sb", C.c_int64), ("x_st", C.c_int64),
                ("o_sb", C.c_int64), ("o_st", C.c_int64),
                ("rowmap", C.c_void_p), ("rowlen", C.c_int32), ("kind", C.c_int32),
                ("alpha", C.c_void_p), ("beta", C.c_void_p), ("C", C.c_int32), ("inner", C.c_int32),
                ("add", C.c_void_p), ("add_st", C.c_int64), ("add_period", C.c_int64),
                ("tau", C.c_float), ("v_th", C.c_float), ("v_reset", C.c_float), ("soft_reset", C.c_int32),
                ("psn_w", C.c_void_p), ("psn_b", C.c_void_p),
                ("nrep", C.c_int64), ("x_srep", C.c_int64), ("o_srep", C.c_int64)]


class SpikeGemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("Wp", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int64), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int64), ("ldo", C.c_int64), ("nsplit", C.c_int32),
                ("bias", C.c_void_p), ("alpha", C.c_void_p), ("beta", C.c_void_p),
                ("resid", C.c_void_p), ("out_rowmap", C.c_void_p),
                ("zg_nH", C.c_int32), ("zg_T", C.c_int32), ("zg_B", C.c_int32), ("zg_N1", C.c_int32),
                ("sn_T", C.c_int32), ("sn_kind", C.c_int32),
                ("tau", C.c_float), ("v_th", C.c_float), ("v_reset", C.c_float), ("soft_reset", C.c_int32),
                ("psn_w", C.c_void_p), ("psn_b", C.c_void_p),
                ("pos_count", C.c_int64), ("pos_inner", C.c_int64), ("pos_ostride", C.c_int64), ("t_stride", C.c_int64),
                ("add", C.c_void_p), ("add_prows", C.c_int64), ("out_spike", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64), ("acc_scale", C.c_float), ("out_rows", C.c_int64),
                ("col_scale", C.c_void_p), ("zg_rep", C.c_int32)]


class SpikeConvDesc(C.Structure):
    _fields_ = [("g", SpikeGemmDesc), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("OH", C.c_int32),
                ("OW", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("sy", C.c_int32), ("sx", C.c_int32),
                ("dy", C.c_int32 * 3), ("dx", C.c_int32 * 3)]


class HeadConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("alpha", C.c_void_p), ("beta", C.c_void_p), ("out", C.c_void_p),
                ("B", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("Cout", C.c_int32),
                ("sn_kind", C.c_int32), ("tau", C.c_float), ("v_th", C.c_float), ("v_reset", C.c_float),
                ("soft_reset", C.c_int32), ("psn_w", C.c_void_p), ("psn_b", C.c_void_p),
                ("x_sb", C.c_int64), ("x_st", C.c_int64), ("x_sy", C.c_int64), ("x_sx", C.c_int64), ("x_sc", C.c_int64 * 4)]


class WinAttnDesc(C.Structure):
    _fields_ = [("mode", C.c_int32), ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
                ("B_", C.c_int32), ("nW", C.c_int32), ("nH", C.c_int32), ("N", C.c_int32), ("hd", C.c_int32),
                ("Tq", C.c_int32), ("N1", C.c_int32),
                ("scale", C.c_void_p), ("bias", C.c_void_p), ("mask", C.c_void_p), ("row_map", C.c_void_p), ("pad_qkv", C.c_void_p)]


class AnnMlpBlockDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("rows", C.c_int64), ("C", C.c_int32), ("Ch", C.c_int32),
                ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float), ("w1", C.c_void_p), ("b1", C.c_void_p),
                ("w2", C.c_void_p), ("b2", C.c_void_p)]


class SpikeDeconvDesc(C.Structure):
    _fields_ = [("spikes", C.c_void_p), ("digits", C.c_void_p), ("cscale", C.c_void_p), ("alpha", C.c_void_p), ("beta", C.c_void_p),
                ("out", C.c_void_p), ("imgs", C.c_int32), ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("Cout", C.c_int32)]


class AnnAttnBlockDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("out", C.c_void_p), ("row_map", C.c_void_p),
                ("B_", C.c_int32), ("nW", C.c_int32), ("nH", C.c_int32), ("N", C.c_int32), ("C", C.c_int32), ("rows", C.c_int64),
                ("ln_w", C.c_void_p), ("ln_b", C.c_void_p), ("ln_eps", C.c_float),
                ("wqkv", C.c_void_p), ("qkv_bias", C.c_void_p), ("scale", C.c_void_p), ("table", C.c_void_p),
                ("wproj", C.c_void_p), ("proj_bias", C.c_void_p)]


class DenseConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("alpha", C.c_void_p), ("beta", C.c_void_p), ("resid", C.c_void_p),
                ("out", C.c_void_p), ("imgs", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("cin_records", C.c_int32),
                ("N", C.c_int32), ("relu", C.c_int32), ("out_f32", C.c_int32), ("x_records", C.c_int32), ("acc_scale", C.c_float)]


class DenseLinearDesc(C.Structure):
    _fields_ = [("a", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("resid", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("gelu", C.c_int32),
                ("cv_H", C.c_int32), ("cv_W", C.c_int32), ("cv_C", C.c_int32), ("cv_stride", C.c_int32), ("cv_OH", C.c_int32),
                ("cv_OW", C.c_int32), ("out_T", C.c_int32), ("acc_scale", C.c_float)]


class LinearDwDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("partial", C.c_void_p),
                ("M", C.c_int64), ("N", C.c_int32), ("K", C.c_int32), ("nsplit", C.c_int32), ("cv_C", C.c_int32), ("cv_Wp", C.c_int32)]


class LinearTrainDesc(C.Structure):
    _fields_ = [("a", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("mode", C.c_int32), ("cv_C", C.c_int32), ("cv_Wp", C.c_int32)]


_lib = None
_PROF = None            # bench.py's per-call profile: a list of (entry point, note, start event, end event) while profile_calls() is active
_NOTE = {}


class _TimedLib:
    """The library with every `*_fwd` entry point bracketed by two HIP events on the stream it launches on (profile_calls())."""

    def __init__(self, cdll):
        self._c = cdll

    def __getattr__(self, name):
        f = getattr(self._c, name)
        if not name.endswith("_fwd"):
            return f

        def timed(*a):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            note = dict(_NOTE)
            _NOTE.clear()
            e0.record()
            rc = f(*a)
            e1.record()
            _PROF.append((name, note, e0, e1))
            return rc
        return timed


def _note(**kw):
    """What the next C-ABI call computes (algorithmic flop / bytes, shape) - kept only while a profile is being taken."""
    if _PROF is not None:
        _NOTE.update(kw)


class profile_calls:
    """`with hip.profile_calls() as rec:` - every C-ABI `*_fwd` call inside is timed with HIP events on its own stream; afterwards
    `rec.rows()` = [(entry point, note dict, microseconds)] in call order (bench.py: `roofline.by_time`)."""

    def __enter__(self):
        global _PROF
        lib()
        _PROF = []
        return self

    def __exit__(self, *exc):
        global _PROF
        torch.cuda.synchronize()
        self._rows = [(n, note, e0.elapsed_time(e1) * 1e3) for n, note, e0, e1 in _PROF]
        _PROF = None
        _NOTE.clear()

    def rows(self):
        return self._rows


class LaunchRecord(C.Structure):
    _fields_ = [("kernel", C.c_char * 192), ("workgroups", C.c_uint32), ("threads", C.c_uint32), ("lds_bytes", C.c_uint32), ("us", C.c_float)]


class launch_log:
    """`with hip.launch_log() as log:` - every kernel launch the library makes inside is timed with HIP events on its own stream
    (sdf_launch_log); afterwards `log.rows` = [(kernel name, workgroups, threads per workgroup, dynamic LDS bytes, microseconds)] in
    launch order.  workgroups x microseconds / 256 = the launch's chip time.  Eager launches only (not under graph capture)."""

    def __enter__(self):
        lib().sdf_launch_log(C.c_int(1))
        return self

    def __exit__(self, *exc):
        c = _lib
        c.sdf_launch_log(C.c_int(0))
        n = c.sdf_launch_log_read(None, C.c_int(0))
        buf = (LaunchRecord * max(n, 1))()
        c.sdf_launch_log_read(buf, C.c_int(n))
        self.rows = [(buf[i].kernel.decode(errors="replace"), int(buf[i].workgroups), int(buf[i].threads), int(buf[i].lds_bytes), float(buf[i].us))
                     for i in range(n)]


def lib():
    """Load the shared library once; fail loudly when it is absent."""
    global _lib
    if _lib is not None and _PROF is not None:
        return _TimedLib(_lib)
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SdfError(f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(sdformerflow_amd/csrc/build.sh); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.sdf_version.restype = C.c_int
        for name in EXPORTS[1:]:
            getattr(_lib, name).restype = None if name in VOID_EXPORTS else C.c_int
        _lib.sdf_psn_bwd_workspace_bytes.restype = C.c_int64
        _lib.sdf_qk_attn_workspace_bytes.restype = C.c_int64
        _lib.sdf_ms_mlp_workspace_bytes.restype = C.c_int64
        _lib.sdf_bn_train_workspace_bytes.restype = C.c_int64
        _lib.sdf_qk_gate_bwd_workspace_bytes.restype = C.c_int64
    return _lib


E_NULL, E_SHAPE, E_DTYPE, E_ALIGN = -1, -2, -3, -4          # include/sdformerflow_hip.h: SDF_E_*


def _check(rc, what):
    if rc != 0:
        kind = "argument error" if rc < 0 else "hipError_t"
        raise SdfError(f"{what} failed: {kind} {rc}", rc=rc)


def _ptr(t, dtype=None):
    if t is None:
        return None
    if not t.is_cuda:
        raise SdfError("HIP path needs device tensors (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise SdfError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_WS = {}


def workspace(device, nbytes=96 << 20):
    """One caller-owned scratch buffer per (device, stream) for the split-K path of the spike matmul / conv kernels:
    forwards running concurrently on different streams must not share partial sums."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    if key not in _WS or _WS[key].numel() < nbytes:
        _WS[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return _WS[key]


def _set_ws(g, like):
    ws = workspace(like.device)
    g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel()


class NeuronParams:
    """Neuron hyper-parameters (the YAML `spiking_neuron` block) plus PSN weights when kind == 'psn'."""

    def __init__(self, kind="lif", tau=2.0, v_th=1.0, v_reset=None, psn_w=None, psn_b=None):
        self.kind, self.tau, self.v_th, self.v_reset = kind, float(tau), float(v_th), v_reset
        self.psn_w, self.psn_b = psn_w, psn_b


def _neuron_desc(x, out, T, nb, ni, x_sb, x_st, o_sb, o_st, p: NeuronParams, rowmap=None, rowlen=0,
                 alpha=None, beta=None, Cch=0, inner=1, add=None, add_st=0, add_period=0, v_last=None, rep=None):
    d = NeuronDesc()
    if rep is not None:
        d.nrep, d.x_srep, d.o_srep = rep                          # (count, stride in x, stride in out) of the outermost dimension
    d.x, d.out, d.v_last = _ptr(x, torch.float32), _ptr(out), _ptr(v_last, torch.float32)
    d.T = T
    d.out_dtype = SDF_F32 if out.dtype == torch.float32 else SDF_U8
    if out.dtype not in (torch.float32, torch.uint8):
        raise SdfError(f"spike dtype {out.dtype} unsupported")
    d.nb, d.ni, d.x_sb, d.x_st, d.o_sb, d.o_st = nb, ni, x_sb, x_st, o_sb, o_st
    d.rowmap, d.rowlen = _ptr(rowmap, torch.int32), rowlen
    d.kind = KIND[p.kind]
    d.alpha, d.beta, d.C, d.inner = _ptr(alpha, torch.float32), _ptr(beta, torch.float32), Cch, inner
    d.add, d.add_st, d.add_period = _ptr(add, torch.float32), add_st, add_period
    d.tau, d.v_th = p.tau, p.v_th
    d.v_reset = 0.0 if p.v_reset is None else float(p.v_reset)
    d.soft_reset = 1 if p.v_reset is None else 0
    d.psn_w, d.psn_b = _ptr(p.psn_w, torch.float32), _ptr(p.psn_b, torch.float32)
    return d


def neuron_fwd(x, out, T, nb, ni, x_sb, x_st, o_sb, o_st, p: NeuronParams, rowmap=None, rowlen=0,
               alpha=None, beta=None, Cch=0, inner=1, add=None, add_st=0, add_period=0, v_last=None):
    """sdf_neuron_fwd: see include/sdformerflow_hip.h for the addressing contract."""
    d = _neuron_desc(x, out, T, nb, ni, x_sb, x_st, o_sb, o_st, p, rowmap, rowlen, alpha, beta, Cch, inner, add, add_st, add_period, v_last)
    _note(bytes=T * nb * ni * (4 + out.element_size()), shape=(T, nb * ni))
    _check(lib().sdf_neuron_fwd(C.byref(d), _stream()), "sdf_neuron_fwd")
    return out


def neuron_multi_fwd(calls):
    """sdf_neuron_multi_fwd: `calls` = argument tuples of neuron_fwd; one launch when they share T and are at most six."""
    descs = (NeuronDesc * len(calls))(*[_neuron_desc(*c) for c in calls])
    _check(lib().sdf_neuron_multi_fwd(descs, C.c_int(len(calls)), _stream()), "sdf_neuron_multi_fwd")


def _pad4(*ts):
    """The streaming neuron kernels move 4 neurons per lane: a per-step size that is not a multiple of 4 (e.g. the
    (T', B_, 81, 3) token gate of the QK attention) is run on a zero-padded (T, N4) copy and sliced back."""
    shape = ts[0].shape
    T, N = shape[0], ts[0][0].numel()
    if N % 4 == 0:
        return [t.contiguous() for t in ts], None
    pad = 4 - N % 4
    return [torch.nn.functional.pad(t.reshape(T, N), (0, pad)) for t in ts], (shape, N)


def _unpad4(out, info):
    return out if info is None else out[:, :info[1]].reshape(info[0])


def lif_fwd(x, tau=2.0, v_th=1.0, v_reset=None, out_dtype=torch.float32, return_v=False):
    """Multi-step LIF over dim 0 of a contiguous tensor (sdf_lif_fwd; any per-step size - the library handles N % 4 != 0)."""
    x = x.contiguous()
    T, N = x.shape[0], x[0].numel()
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    v = torch.empty(x.shape[1:], dtype=torch.float32, device=x.device) if return_v else None
    rc = lib().sdf_lif_fwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(out)), C.c_void_p(_ptr(v)),
                           C.c_int(T), C.c_int64(N), C.c_float(tau), C.c_float(v_th),
                           C.c_int(1 if v_reset is None else 0), C.c_float(0.0 if v_reset is None else v_reset),
                           C.c_int(SDF_F32 if out_dtype == torch.float32 else SDF_U8), _stream())
    _check(rc, "sdf_lif_fwd")
    return (out, v) if return_v else out


def lif_bwd(x, grad_spike, tau=2.0, v_th=1.0, v_reset=None, detach_reset=True, alpha=2.0, kind="lif"):
    """BPTT through the multi-step LIF / IF over dim 0 (sdf_lif_bwd): dL/dx from x and dL/dspike; ATan surrogate."""
    (x, g), info = _pad4(x, grad_spike)
    if info is not None:
        return _unpad4(lif_bwd(x, g, tau, v_th, v_reset, detach_reset, alpha, kind), info)
    T, N = x.shape[0], x[0].numel()
    gx = torch.empty_like(x)
    rc = lib().sdf_lif_bwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(g, torch.float32)), C.c_void_p(_ptr(gx)),
                           C.c_int(T), C.c_int64(N), C.c_int(KIND[kind]), C.c_float(tau), C.c_float(v_th),
                           C.c_int(1 if v_reset is None else 0), C.c_float(0.0 if v_reset is None else v_reset),
                           C.c_int(1 if detach_reset else 0), C.c_int(0), C.c_float(alpha), _stream())
    _check(rc, "sdf_lif_bwd")
    return gx


def psn_bwd(x, W, b, grad_spike, alpha=2.0, need_param_grads=True):
    """Backward of the parallel spiking neuron (sdf_psn_bwd): (dL/dx, dL/dW, dL/db); ATan surrogate.
    T <= 10 reduces dW / db in the kernel; larger T takes grad_h from the kernel and one library GEMM."""
    (x, g), info = _pad4(x, grad_spike)
    if info is not None:                     # padded columns carry x = 0, dL/ds = 0: they add nothing to dW / db
        gx, gW, gb = psn_bwd(x, W, b, g, alpha, need_param_grads)
        return _unpad4(gx, info), gW, gb
    T, N = x.shape[0], x[0].numel()
    gx = torch.empty_like(x)
    fused = need_param_grads and T <= 10
    gW = torch.empty((T, T), dtype=torch.float32, device=x.device) if fused else None
    gb = torch.empty((T,), dtype=torch.float32, device=x.device) if fused else None
    gh = torch.empty_like(x) if (need_param_grads and not fused) else None
    nbytes = lib().sdf_psn_bwd_workspace_bytes(C.c_int(T), C.c_int64(N)) if fused else 0
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=torch.float32, device=x.device) if fused else None
    rc = lib().sdf_psn_bwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(W.contiguous(), torch.float32)),
                           C.c_void_p(_ptr(b.contiguous().view(-1), torch.float32)), C.c_void_p(_ptr(g, torch.float32)),
                           C.c_void_p(_ptr(gx)), C.c_void_p(_ptr(gW)), C.c_void_p(_ptr(gb)), C.c_void_p(_ptr(gh)),
                           C.c_void_p(_ptr(ws)), C.c_int64(nbytes), C.c_int(T), C.c_int64(N), C.c_int(0), C.c_float(alpha),
                           _stream())
    _check(rc, "sdf_psn_bwd")
    if gh is not None:
        gW, gb = gh.view(T, -1) @ x.view(T, -1).t(), gh.view(T, -1).sum(1)
    return gx, gW, gb


def psn_fwd(x, W, b, out_dtype=torch.float32):
    """Parallel spiking neuron over dim 0 of a contiguous tensor (sdf_psn_fwd; any per-step size)."""
    x = x.contiguous()
    T, N = x.shape[0], x[0].numel()
    out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    rc = lib().sdf_psn_fwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(W.contiguous(), torch.float32)),
                           C.c_void_p(_ptr(b.contiguous().view(-1), torch.float32)), C.c_void_p(_ptr(out)),
                           C.c_int(T), C.c_int64(N), C.c_int(SDF_F32 if out_dtype == torch.float32 else SDF_U8),
                           _stream())
    _check(rc, "sdf_psn_fwd")
    return out


def split_weight(W, nsplit=3):
    """fp32 weight (N,K) -> 16-bit planes (nsplit,N,K) stored as int16 bit patterns.
    nsplit 1 / 3: bf16 planes (sdf_split_weight_bf16).  nsplit 2: two fp16 planes of scale * W
    (sdf_split_weight_f16x2) with scale the power of two that puts max|W| in [2^14, 2^15); the matching accumulator
    scale 1 / scale travels with the tensor as the attribute `sdf_acc_scale`."""
    W = W.contiguous()
    planes = torch.empty((nsplit,) + tuple(W.shape), dtype=torch.int16, device=W.device)
    if nsplit == 2:
        import math
        mx = float(W.detach().abs().max())
        scale = 2.0 ** (14 - math.floor(math.log2(mx))) if mx > 0 and math.isfinite(mx) else 1.0
        scale = min(max(scale, 2.0 ** -100), 2.0 ** 100)
        _check(lib().sdf_split_weight_f16x2(C.c_void_p(_ptr(W, torch.float32)), C.c_void_p(planes.data_ptr()),
                                            C.c_int64(W.numel()), C.c_float(scale), _stream()), "sdf_split_weight_f16x2")
        planes.sdf_acc_scale = 1.0 / scale
        return planes
    _check(lib().sdf_split_weight_bf16(C.c_void_p(_ptr(W, torch.float32)), C.c_void_p(planes.data_ptr()),
                                       C.c_int64(W.numel()), C.c_int(nsplit), _stream()), "sdf_split_weight_bf16")
    return planes


PLANES_I8X3 = 4
PLANES_I8X3_TILED = 5


def split_weight_i8x3(W):
    """fp32 weight (N, K) -> int8 digit planes (3, N, K) + per-row power-of-two scales (attribute `sdf_col_scale`):
    w = (d2*65536 + d1*256 + d0) * scale (sdf_split_weight_i8x3).  Read by the weight-resident 3x3 convolution and the wide-stage
    kernels of the swin blocks (csrc/ms_wide.hip)."""
    W = W.contiguous()
    N, K = W.shape
    planes = torch.empty((3, N, K), dtype=torch.int8, device=W.device)
    scale = torch.empty((N,), dtype=torch.float32, device=W.device)
    _check(lib().sdf_split_weight_i8x3(C.c_void_p(_ptr(W, torch.float32)), C.c_void_p(planes.data_ptr()), C.c_void_p(scale.data_ptr()),
                                       C.c_int(N), C.c_int(K), _stream()), "sdf_split_weight_i8x3")
    planes.sdf_col_scale = scale
    return planes


def conv_wres_applicable(imgs, H, W, Cin, Cout, stride, T=1):
    """Mirror of the library's dispatch rule (csrc/spike_conv_wres.hip: spike_conv_wres_supports): 3x3 on 96 input channels at
    stride 1 or - digit planes only - at stride 2 on 48 (the patch embedding's first 3x3) or 96 channels (its projection), output columns in blocks of 32, and enough
    8 x 16 pixel OUTPUT tiles (x T steps each when the neuron is fused) to give every half workgroup of the chip work - below that
    the streaming kernels' split-K wins.  H, W: the input image."""
    if sw("SDF_CONV_WRES", "") == "0":             # the library's A/B override: always the streaming kernels (which
        return False                                             # do not read digit planes: the caller must keep its 16-bit planes)
    if (Cin, stride) not in ((96, 1), (48, 2), (96, 2)) or Cout % 32 or ((Cin, stride) == (96, 2) and T != 1):
        return False                                             # (96 channels at stride 2: two channel passes, fp32 epilogue only)
    oh, ow = (H - 1) // stride + 1, (W - 1) // stride + 1
    if imgs * max(oh * ow * Cout * 4, H * W * Cin) >= 1 << 31:   # the kernel addresses its operands with 31-bit byte offsets
        return False
    tiles = (imgs // T) * (-(-oh // 8)) * (-(-ow // 16))
    return tiles * (Cout // 32) >= 512


def wide_conv_applicable(imgs, H, W, Cin, Cout, stride, T):
    """Mirror of the library's dispatch rule for the small-M digit convolution (csrc/ms_wide.hip: wide_conv_supports): 3x3 / stride 1
    on Cin % 128 == 0 channels, at most 131 072 output rows (SDF_WIDE_MAXROWS) in (B, T, H, W) order with T in {10, 20}."""
    if sw("SDF_WIDE", "") == "0" or sw("SDF_WIDE_CONV", "") != "1" or stride != 1 or Cin % 128 or Cout % 32:
        return False                                             # (opt-in: measured no faster than the streaming kernel + split-K, docs/history/DESIGN_rounds1-5.md)
    return T in (10, 20) and imgs % T == 0 and imgs * H * W <= int(sw("SDF_WIDE_MAXROWS", 131072))


def smallm_conv_applicable(imgs, H, W, Cin, Cout, stride, T):
    """Mirror of the library's dispatch rule for the small-M digit convolution (csrc/ms_smallm.hip: smallm_conv_supports): 3x3 / stride 1
    on Cin % 64 == 0 channels, at most 32 000 output rows (SDF_SMALLM_CONV_ROWS; 5 120 until round 5) in (B, T, H, W) order with T in {10, 20}."""
    if sw("SDF_SMALLM", "") == "0" or stride != 1 or Cin % 64 or Cout % 32:
        return False
    return T in (10, 20) and imgs % T == 0 and imgs * H * W <= int(sw("SDF_SMALLM_CONV_ROWS", 400 * 80))


def tile_weight_i8x3(planes):
    """sdf_tile_weight_i8x3: digit planes (3, N, K) -> the same digits in MFMA fragment order [N / 16][K / 64][3][64][16 B] (kept under
    the shape (3, N, K); attribute `sdf_tiled`): what the small-M convolution streams at full rate (csrc/ms_smallm.hip)."""
    _, N, K = planes.shape
    tiled = torch.empty_like(planes)
    _check(lib().sdf_tile_weight_i8x3(C.c_void_p(_ptr(planes, torch.int8)), C.c_void_p(tiled.data_ptr()), C.c_int(N), C.c_int(K), _stream()),
           "sdf_tile_weight_i8x3")
    tiled.sdf_col_scale = planes.sdf_col_scale
    tiled.sdf_tiled = True
    return tiled


def smallm_gemm_applicable(M, N, K):
    """Mirror of the library's rule for the plain small-M product (csrc/ms_smallm.hip: smallm_gemm_supports) plus the engine's
    own bound: it pays where K is long enough to amortise a workgroup's prologue and reduction (measured: the first decoder's
    1 080 x 3 456 x 1 536 product; not the second's 4 320 x 1 728 x 832)."""
    if sw("SDF_SMALLM", "") == "0" or M % 10 or N % 32 or K % 64:
        return False
    return M <= 64 * 80 and K >= int(sw("SDF_SMALLM_MINK", 1024))


def smallm_gemm_rows_ok(M):
    """Rows ONE launch of the plain small-M product admits (the library's bound, csrc/ms_smallm.hip smallm_gemm_supports) - wider than the
    routing rule above, which decides per sample: replicas run the product their batch-1 forward takes, in as few launches as this allows."""
    return M % 10 == 0 and M <= int(sw("SDF_SMALLM_CONV_ROWS", 400 * 80))


def res_gemm_applicable(M, N, K):
    """Mirror of the library's rule for the plain product on row-major digit planes (csrc/spike_gemm.hip -> ms_res.hip: whole-K digits
    LDS-resident per 32 columns, row loop): rows in tens, K <= 1024 in steps of 16."""
    if sw("SDF_RES", "") == "0" or sw("SDF_RES_GEMM", "") == "0":
        return False
    return M % 10 == 0 and N % 32 == 0 and K % 16 == 0 and 32 <= K <= 1024 and M * max(K, 4 * N) < 1 << 31


def pack_conv_weight_i8x3(w, tiled=False):
    """Conv2d weight (Cout, Cin, KH, KW) fp32 -> int8 digit planes (3, Cout, KH*KW*Cin), K in (ky, kx, cin) order; `tiled`: in the
    fragment order of tile_weight_i8x3."""
    Cout, Cin, KH, KW = w.shape
    planes = split_weight_i8x3(w.detach().float().permute(0, 2, 3, 1).reshape(Cout, KH * KW * Cin))
    return tile_weight_i8x3(planes) if tiled else planes


def _acc_scale(Wp):
    if Wp.dtype == torch.int8 or Wp.shape[0] != 2:
        return 0.0
    if not hasattr(Wp, "sdf_acc_scale"):
        raise SdfError("fp16 weight planes without their scale: use the tensor returned by split_weight(W, 2) as is")
    return Wp.sdf_acc_scale


def spike_gemm(A, Wp, out, M, N, K, lda=None, ldo=None, bias=None, alpha=None, beta=None, resid=None,
               out_rowmap=None, zg=None):
    """sdf_spike_gemm_fwd.  A: u8 spikes, Wp: int16 (nsplit,N,K) bf16 planes, out: fp32.
    zg = (nH, Tq, B_, N1[, windows per replica]) selects the head-scramble A addressing."""
    d = SpikeGemmDesc()
    d.A, d.Wp, d.out = _ptr(A, torch.uint8), _ptr(Wp), _ptr(out, torch.float32)
    d.M, d.N, d.K = M, N, K
    d.lda = K if lda is None else lda
    d.ldo = N if ldo is None else ldo
    d.nsplit, d.acc_scale = Wp.shape[0], _acc_scale(Wp)
    if Wp.dtype == torch.int8:
        # digit planes: in fragment order (tile_weight_i8x3) the small-M kernel's operand, row-major (split_weight_i8x3) the
        # weight-resident row-loop kernel's (csrc/ms_res.hip: K <= 1024, M % 10 == 0)
        d.nsplit = PLANES_I8X3_TILED if getattr(Wp, "sdf_tiled", False) else PLANES_I8X3
        d.col_scale = _ptr(Wp.sdf_col_scale, torch.float32)
    elif Wp.dtype != torch.int16:
        raise SdfError(f"weight planes must be int16 (16-bit float planes) or int8 (tiled digit planes), got {Wp.dtype}")
    d.bias, d.alpha, d.beta = _ptr(bias, torch.float32), _ptr(alpha, torch.float32), _ptr(beta, torch.float32)
    d.resid, d.out_rowmap = _ptr(resid, torch.float32), _ptr(out_rowmap, torch.int32)
    if out_rowmap is not None:
        d.out_rows = out.numel() // d.ldo
    if zg is not None:
        d.zg_nH, d.zg_T, d.zg_B, d.zg_N1 = zg[:4]
        d.zg_rep = zg[4] if len(zg) > 4 else 0                 # (windows per independent replica: SdfSpikeGemmDesc.zg_rep)
    _set_ws(d, A)
    _note(flop=2 * int(d.M) * int(d.N) * int(d.K), shape=(int(d.M), int(d.N), int(d.K)))
    _check(lib().sdf_spike_gemm_fwd(C.byref(d), _stream()), "sdf_spike_gemm_fwd")
    return out


def spike_gemm_sn(A, Wp, out_spike, N, K, T, pos_count, pos_inner, pos_ostride, t_stride, p: NeuronParams,
                  alpha=None, beta=None, add=None, add_prows=0, lda=None):
    """sdf_spike_gemm_fwd with the fused neuron epilogue: out_spike (rows,N) u8 = SN_T(BN(A W^T) [+ add])."""
    d = SpikeGemmDesc()
    d.A, d.Wp, d.out_spike = _ptr(A, torch.uint8), _ptr(Wp, torch.int16), _ptr(out_spike, torch.uint8)
    d.M, d.N, d.K = pos_count * T, N, K
    d.lda, d.ldo, d.nsplit, d.acc_scale = (K if lda is None else lda), N, Wp.shape[0], _acc_scale(Wp)
    d.alpha, d.beta = _ptr(alpha, torch.float32), _ptr(beta, torch.float32)
    d.sn_T, d.sn_kind, d.tau, d.v_th = T, KIND[p.kind], p.tau, p.v_th
    d.v_reset, d.soft_reset = (0.0 if p.v_reset is None else float(p.v_reset)), (1 if p.v_reset is None else 0)
    d.psn_w, d.psn_b = _ptr(p.psn_w, torch.float32), _ptr(p.psn_b, torch.float32)
    d.pos_count, d.pos_inner, d.pos_ostride, d.t_stride = pos_count, pos_inner, pos_ostride, t_stride
    d.add, d.add_prows = _ptr(add, torch.float32), add_prows
    _note(flop=2 * int(d.M) * int(d.N) * int(d.K), shape=(int(d.M), int(d.N), int(d.K)))
    _check(lib().sdf_spike_gemm_fwd(C.byref(d), _stream()), "sdf_spike_gemm_fwd")
    return out_spike


def qk_gate(q, k, e, Tq, rows, Cch, p: NeuronParams, ldq=None, ldk=None):
    """sdf_qk_gate_strided_fwd on u8 spike tensors laid out (Tq, rows, C); q / k rows may be ldq / ldk bytes apart
    (the halves of a fused q|k GEMM output)."""
    rc = lib().sdf_qk_gate_strided_fwd(C.c_void_p(_ptr(q, torch.uint8)), C.c_void_p(_ptr(k, torch.uint8)),
                               C.c_void_p(_ptr(e, torch.uint8)), C.c_int(Tq), C.c_int64(rows), C.c_int(Cch),
                               C.c_int64(Cch if ldq is None else ldq), C.c_int64(Cch if ldk is None else ldk),
                               C.c_int(KIND[p.kind]), C.c_float(p.tau), C.c_float(p.v_th),
                               C.c_float(0.0 if p.v_reset is None else p.v_reset),
                               C.c_int(1 if p.v_reset is None else 0),
                               C.c_void_p(_ptr(p.psn_w, torch.float32)), C.c_void_p(_ptr(p.psn_b, torch.float32)),
                               _stream())
    _check(rc, "sdf_qk_gate_strided_fwd")
    return e


class NeuronCfg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("tau", C.c_float), ("v_th", C.c_float), ("v_reset", C.c_float),
                ("soft_reset", C.c_int32), ("psn_w", C.c_void_p), ("psn_b", C.c_void_p)]


class QkAttnDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("slice_map", C.c_void_p), ("B_", C.c_int64), ("x_rows", C.c_int64),
                ("Tq", C.c_int32), ("N1", C.c_int32), ("C", C.c_int32), ("nH", C.c_int32), ("nsplit", C.c_int32),
                ("qk_planes", C.c_void_p), ("qk_alpha", C.c_void_p), ("qk_beta", C.c_void_p), ("qk_add", C.c_void_p),
                ("qk_acc_scale", C.c_float),
                ("q_planes", C.c_void_p), ("q_alpha", C.c_void_p), ("q_beta", C.c_void_p), ("q_acc_scale", C.c_float),
                ("k_planes", C.c_void_p), ("k_alpha", C.c_void_p), ("k_beta", C.c_void_p), ("k_add", C.c_void_p),
                ("k_acc_scale", C.c_float),
                ("p_planes", C.c_void_p), ("p_bias", C.c_void_p), ("p_alpha", C.c_void_p), ("p_beta", C.c_void_p),
                ("p_acc_scale", C.c_float),
                ("sn_proj", NeuronCfg), ("sn_q", NeuronCfg), ("sn_k", NeuronCfg), ("sn2_q", NeuronCfg),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("gemm_workspace", C.c_void_p), ("gemm_workspace_bytes", C.c_int64), ("flags", C.c_int32),
                ("x_src", C.c_void_p), ("xB", C.c_int32), ("xD", C.c_int32), ("xHW", C.c_int64), ("emit_s1", C.c_void_p),
                ("emit_sn", NeuronCfg),
                ("qk_digits", C.c_void_p), ("qk_cscale", C.c_void_p), ("q_digits", C.c_void_p), ("q_cscale", C.c_void_p),
                ("k_digits", C.c_void_p), ("k_cscale", C.c_void_p), ("p_digits", C.c_void_p), ("p_cscale", C.c_void_p),
                ("rep_windows", C.c_int32)]


SDF_QK_KEEP_SPIKES, SDF_QK_FOUR_LAUNCHES, SDF_QK_NARROW = 1, 2, 4


def _ncfg(c, p: NeuronParams):
    c.kind, c.tau, c.v_th = KIND[p.kind], p.tau, p.v_th
    c.v_reset, c.soft_reset = (0.0 if p.v_reset is None else float(p.v_reset)), (1 if p.v_reset is None else 0)
    c.psn_w, c.psn_b = _ptr(p.psn_w, torch.float32), _ptr(p.psn_b, torch.float32)


def window_slice_map(B, D, H, W, ws, ss, device):
    """sdf_window_slice_map: the int32 gather table of pad + roll + window_partition_v2, built on the device.
    Returns (map (B_*Wd*Wh*Ww,), B_)."""
    Wd, Wh, Ww = ws
    B_ = B * -(-D // Wd) * -(-H // Wh) * -(-W // Ww)
    m = torch.empty((B_ * Wd * Wh * Ww,), dtype=torch.int32, device=device)
    nw = C.c_int64(0)
    _check(lib().sdf_window_slice_map(C.c_void_p(m.data_ptr()), C.c_int(B), C.c_int(D), C.c_int(H), C.c_int(W), C.c_int(Wd),
                                      C.c_int(Wh), C.c_int(Ww), C.c_int(ss[0]), C.c_int(ss[1]), C.c_int(ss[2]), C.byref(nw),
                                      _stream()), "sdf_window_slice_map")
    assert nw.value == B_
    return m, B_


def replica_slice_map(map1, nW, R, Tq, N1, x_rows1):
    """The slice map of R INDEPENDENT batch-1 problems from the batch-1 table `map1` ((Tq * nW * N1,) int32, nW windows, x_rows1
    rows of x per sample): entry (t' * R * nW + r * nW + b') * N1 + n = r * x_rows1 + map1[(t' * nW + b') * N1 + n] (-1 stays -1).
    Attention step t' of window b' of replica r is then that replica's own slice t' * nW + b' - the reference's batch-1
    `window_partition_v2` view (Spiking_swin_transformer3D.py:100-113) per sample instead of its batch-coupling view over R samples -
    while the kernels see ONE (Tq, R * nW, N1, C) problem.  Index arithmetic on the device, no kernel."""
    m = map1.view(Tq, 1, nW * N1).expand(Tq, R, nW * N1)
    off = (torch.arange(R, device=map1.device, dtype=torch.int32) * x_rows1).view(1, R, 1)
    return torch.where(m >= 0, m + off, m).reshape(-1).contiguous()


def replica_zsrc_map(z1, nW, R, Tq, N1, Cc):
    """The projection's operand map (window_zsrc_map) of R independent batch-1 problems from the batch-1 table `z1` (int32 per row of
    x: byte offset of the row's gated spikes in the replica's OWN E_flat, a (Tq, nW, N1, C) tensor): that tensor is the sub-block
    [:, r * nW : (r + 1) * nW] of the (Tq, R * nW, N1, C) buffer the kernels share, so offset o of replica r moves to
    ((o // half) * R * nW + r * nW) * N1 * C + o % half with half = nW * N1 * C.  nW % Tq == 0 keeps the nH pieces of a row on one side
    of `half` (the kernels step from head to head by a constant Tq * N1 * 32)."""
    if nW % Tq:
        raise ReplicaGeometryError(f"replicas need a window count per sample ({nW}) that is a multiple of the window depth ({Tq})")
    half = nW * N1 * Cc
    te = z1 // half
    r = torch.arange(R, device=z1.device, dtype=torch.int32).view(R, 1)
    out = (te.view(1, -1) * (R * nW) + r * nW) * (N1 * Cc) + (z1 - te * half).view(1, -1)
    if R * Tq * nW * N1 * Cc >= 1 << 31:
        raise SdfError("replica batch too large for the 31-bit operand offsets of the projection")
    return out.to(torch.int32).reshape(-1).contiguous()


def _digits(dg):
    """(planes pointer, channel-scale pointer) of int8 digit planes made by split_weight_i8x3, or (None, None)."""
    if dg is None:
        return None, None
    return _ptr(dg, torch.int8), _ptr(dg.sdf_col_scale, torch.float32)


def window_zsrc_map(slice_map, B_, Tq, N1, nH, x_rows):
    """sdf_window_zsrc_map: per row of x the byte offset of its gated spikes in E behind the reference's head scramble - what the
    wide-stage projection (csrc/ms_wide.hip) gathers its operand through.  int32 (x_rows,)."""
    z = torch.zeros((x_rows,), dtype=torch.int32, device=slice_map.device)
    _check(lib().sdf_window_zsrc_map(C.c_void_p(slice_map.data_ptr()), C.c_int64(B_), C.c_int(Tq), C.c_int(N1), C.c_int(nH),
                                     C.c_void_p(z.data_ptr()), _stream()), "sdf_window_zsrc_map")
    return z


def qk_attn(x, slice_map, B_, Tq, N1, nH, p_lin, sn_proj, sn_q, sn_k, sn2_q, qk=None, q_lin=None, k_lin=None, pe=None, keep_ws=None,
            four_launches=False, x_src=None, emit=None, narrow=False, info=None, rep_windows=0):
    """sdf_qk_attn_fwd: x (B,D,H,W,C) fp32 channel-last += SSA(x), in place.  `qk` = {"Wp", "alpha", "beta", "add"} for the
    stacked projection, or q_lin / k_lin (objects with Wp / alpha / beta) + pe for separate ones; p_lin has Wp / bias / alpha / beta.
    `x_src` (window_zsrc_map) lets the library take its wide-stage kernels; `emit` = (u8 buffer, NeuronParams): the projection
    also writes SN(x after the update) over D there (the MLP's first neuron) when the call runs on the wide-stage kernels - `info`
    (a dict) then receives "emitted": True; `narrow` forces the general kernels (A/B).  `rep_windows` > 0: the B_ windows are
    B_ / rep_windows independent batch-1 problems and `slice_map` / `x_src` their concatenated tables (replica_slice_map, replica_zsrc_map)."""
    Cc = x.shape[-1]
    d = QkAttnDesc()
    d.rep_windows = rep_windows
    d.x, d.slice_map = _ptr(x, torch.float32), _ptr(slice_map, torch.int32)
    d.B_, d.x_rows, d.Tq, d.N1, d.C, d.nH = B_, x.numel() // Cc, Tq, N1, Cc, nH
    d.nsplit = p_lin.Wp.shape[0]
    if qk is not None:
        d.qk_planes, d.qk_alpha, d.qk_beta = _ptr(qk["Wp"], torch.int16), _ptr(qk["alpha"], torch.float32), _ptr(qk["beta"], torch.float32)
        d.qk_add, d.qk_acc_scale = _ptr(qk["add"], torch.float32), _acc_scale(qk["Wp"])
    else:
        d.q_planes, d.q_alpha, d.q_beta, d.q_acc_scale = _ptr(q_lin.Wp, torch.int16), _ptr(q_lin.alpha), _ptr(q_lin.beta), _acc_scale(q_lin.Wp)
        d.k_planes, d.k_alpha, d.k_beta, d.k_acc_scale = _ptr(k_lin.Wp, torch.int16), _ptr(k_lin.alpha), _ptr(k_lin.beta), _acc_scale(k_lin.Wp)
        d.k_add = _ptr(pe, torch.float32)
    d.p_planes, d.p_bias, d.p_alpha, d.p_beta = _ptr(p_lin.Wp, torch.int16), _ptr(p_lin.bias), _ptr(p_lin.alpha), _ptr(p_lin.beta)
    d.p_acc_scale = _acc_scale(p_lin.Wp)
    for c, p in ((d.sn_proj, sn_proj), (d.sn_q, sn_q), (d.sn_k, sn_k), (d.sn2_q, sn2_q)):
        _ncfg(c, p)
    nbytes = lib().sdf_qk_attn_workspace_bytes(C.c_int64(B_), C.c_int(Tq), C.c_int(N1), C.c_int(Cc))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)          # intermediates: caller-owned, caching allocator
    d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
    gws = workspace(x.device)
    d.gemm_workspace, d.gemm_workspace_bytes = gws.data_ptr(), gws.numel()
    d.flags = (SDF_QK_KEEP_SPIKES if keep_ws is not None else 0) | (SDF_QK_FOUR_LAUNCHES if four_launches else 0) | \
        (SDF_QK_NARROW if narrow else 0)
    if x_src is not None:
        d.x_src, d.xB, d.xD, d.xHW = _ptr(x_src, torch.int32), x.shape[0], x.shape[1], x.shape[2] * x.shape[3]
        # the wide-stage kernels multiply int8 digit planes (`digits` of the layer objects / of the qk dict, made by split_weight_i8x3)
        d.p_digits, d.p_cscale = _digits(getattr(p_lin, "digits", None))
        if qk is not None:
            d.qk_digits, d.qk_cscale = _digits(qk.get("digits"))
        else:
            d.q_digits, d.q_cscale = _digits(getattr(q_lin, "digits", None))
            d.k_digits, d.k_cscale = _digits(getattr(k_lin, "digits", None))
    if emit is not None:
        # the query sees the descriptor as the call will (emit neuron included: the library refuses neurons it has no epilogue for);
        # refused -> the call runs without the emission and the caller's MLP runs its own first neuron
        d.emit_s1 = _ptr(emit[0], torch.uint8)                # (only the wide-stage projection emits the next neuron's spikes)
        _ncfg(d.emit_sn, emit[1])
        if lib().sdf_qk_attn_is_wide(C.byref(d)) == 1:
            if info is not None:
                info["emitted"] = True
        else:
            d.emit_s1 = None
            _ncfg(d.emit_sn, NeuronParams())
    _note(flop=2 * Tq * B_ * N1 * Cc * 3 * Cc, shape=(Tq * B_ * N1, Cc))
    _check(lib().sdf_qk_attn_fwd(C.byref(d), _stream()), "sdf_qk_attn_fwd")
    if keep_ws is not None:
        keep_ws.append(ws)                  # the call's intermediates (u8 spikes, layout: sdf_qk_attn_workspace_bytes) for the parity tape
    return x


class MsMlpDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("B", C.c_int32), ("D", C.c_int32), ("HW", C.c_int64), ("C", C.c_int32), ("Ch", C.c_int32),
                ("nsplit", C.c_int32),
                ("fc1_planes", C.c_void_p), ("fc1_alpha", C.c_void_p), ("fc1_beta", C.c_void_p), ("fc1_acc_scale", C.c_float),
                ("fc2_planes", C.c_void_p), ("fc2_alpha", C.c_void_p), ("fc2_beta", C.c_void_p), ("fc2_acc_scale", C.c_float),
                ("sn1", NeuronCfg), ("sn2", NeuronCfg),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("gemm_workspace", C.c_void_p), ("gemm_workspace_bytes", C.c_int64), ("flags", C.c_int32), ("s1_in", C.c_void_p),
                ("fc1_digits", C.c_void_p), ("fc1_cscale", C.c_void_p), ("fc2_digits", C.c_void_p), ("fc2_cscale", C.c_void_p),
                ("emit_next", C.c_void_p), ("emit_sn", NeuronCfg), ("fc2_tiled", C.c_void_p)]


class MsMergeDesc(C.Structure):
    _fields_ = [("spikes", C.c_void_p), ("digits", C.c_void_p), ("cscale", C.c_void_p), ("alpha", C.c_void_p), ("beta", C.c_void_p),
                ("out", C.c_void_p), ("B", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32),
                ("N", C.c_int32)]


MLP_KEEP_SPIKES, MLP_THREE_LAUNCHES, MLP_NARROW = 1, 2, 4


def ms_mlp_workspace(x, Ch):
    """The caller-owned workspace of ms_mlp for x (B,D,H,W,C): u8, SN1's spikes [tokens][C] at its head (what qk_attn's `emit` fills)."""
    B, D, H, W, Cc = x.shape
    nbytes = lib().sdf_ms_mlp_workspace_bytes(C.c_int64(B * D * H * W), C.c_int(Cc), C.c_int(Ch))
    return torch.empty((nbytes,), dtype=torch.uint8, device=x.device)


def ms_mlp(x, fc1, fc2, sn1, sn2, keep_ws=None, three_launches=False, ws=None, s1_ready=False, narrow=False, emit_next=None):
    """sdf_ms_mlp_fwd: x (B,D,H,W,C) fp32 channel-last += MLP(x) over the time axis D, in place.  `keep_ws` (a list) receives
    the workspace with the SN1 / SN2 spikes (parity tape); `three_launches` selects the unfused A/B reference.  `ws` =
    ms_mlp_workspace(...) of the caller; with `s1_ready` its head already holds SN1(x) (wide stages: written by qk_attn's `emit`).
    `emit_next` = (u8 (B,D,H,W,C) buffer, neuron): wide stages only - the last launch also writes that neuron's spikes of the
    updated x (the patch merging's / the bottleneck's first neuron)."""
    B, D, H, W, Cc = x.shape
    d = MsMlpDesc()
    d.x, d.B, d.D, d.HW, d.C, d.Ch, d.nsplit = _ptr(x, torch.float32), B, D, H * W, Cc, fc1.N, fc1.Wp.shape[0]
    d.fc1_planes, d.fc1_alpha, d.fc1_beta, d.fc1_acc_scale = _ptr(fc1.Wp, torch.int16), _ptr(fc1.alpha), _ptr(fc1.beta), _acc_scale(fc1.Wp)
    d.fc2_planes, d.fc2_alpha, d.fc2_beta, d.fc2_acc_scale = _ptr(fc2.Wp, torch.int16), _ptr(fc2.alpha), _ptr(fc2.beta), _acc_scale(fc2.Wp)
    _ncfg(d.sn1, sn1)
    _ncfg(d.sn2, sn2)
    if ws is None:
        if s1_ready:
            raise SdfError("s1_ready needs the caller's workspace (ms_mlp_workspace) with SN1's spikes at its head")
        ws = ms_mlp_workspace(x, fc1.N)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
    gws = workspace(x.device)
    d.gemm_workspace, d.gemm_workspace_bytes = gws.data_ptr(), gws.numel()
    d.flags = (MLP_KEEP_SPIKES if keep_ws is not None else 0) | (MLP_THREE_LAUNCHES if three_launches else 0) | (MLP_NARROW if narrow else 0)
    d.fc1_digits, d.fc1_cscale = _digits(getattr(fc1, "digits", None))
    d.fc2_digits, d.fc2_cscale = _digits(getattr(fc2, "digits", None))
    d.fc2_tiled = _ptr(getattr(fc2, "digits_tiled", None), torch.int8)      # (fragment order: fc2 on the small-M kernel, csrc/ms_smallm.hip)
    if s1_ready:
        d.s1_in = ws.data_ptr()
    if emit_next is not None:
        buf, nsn = emit_next
        if buf.numel() != x.numel() or buf.dtype != torch.uint8:
            raise SdfError("emit_next: need a u8 buffer of x's shape")
        d.emit_next = _ptr(buf, torch.uint8)
        _ncfg(d.emit_sn, nsn)
    if (s1_ready or emit_next is not None) and lib().sdf_ms_mlp_is_wide(C.byref(d)) != 1:
        raise SdfError("s1_ready / emit_next: this MLP does not run on the wide-stage kernels (they would be ignored)")
    _note(flop=4 * B * D * H * W * Cc * fc1.N, shape=(B * D * H * W, Cc, fc1.N))
    _check(lib().sdf_ms_mlp_fwd(C.byref(d), _stream()), "sdf_ms_mlp_fwd")
    if keep_ws is not None:
        keep_ws.append(ws)
    return x


def ms_mlp_is_wide(x, fc1, fc2, sn1, sn2, emit_sn=None):
    """Whether ms_mlp of this layer on x runs on the wide-stage kernels (host-only query; then `s1_ready` / `emit_next` - with the
    neuron `emit_sn` - are accepted)."""
    if getattr(fc1, "digits", None) is None or getattr(fc2, "digits", None) is None:
        return False
    B, D, H, W, Cc = x.shape
    d = MsMlpDesc()
    d.x, d.B, d.D, d.HW, d.C, d.Ch, d.nsplit = _ptr(x, torch.float32), B, D, H * W, Cc, fc1.N, fc1.Wp.shape[0]
    d.fc1_planes, d.fc2_planes = _ptr(fc1.Wp, torch.int16), _ptr(fc2.Wp, torch.int16)
    d.fc1_alpha, d.fc1_beta, d.fc2_alpha, d.fc2_beta = _ptr(fc1.alpha), _ptr(fc1.beta), _ptr(fc2.alpha), _ptr(fc2.beta)
    _ncfg(d.sn1, sn1)
    _ncfg(d.sn2, sn2)
    d.workspace, d.workspace_bytes = 256, 1 << 40               # not dereferenced by the query
    if emit_sn is not None:
        d.emit_next = 256
        _ncfg(d.emit_sn, emit_sn)
    d.fc1_digits, d.fc1_cscale = _digits(fc1.digits)
    d.fc2_digits, d.fc2_cscale = _digits(fc2.digits)
    return lib().sdf_ms_mlp_is_wide(C.byref(d)) == 1


def ms_patch_merge(spikes, lin, out=None):
    """sdf_ms_patch_merge_fwd: spikes u8 (B,D,H,W,C) = SN(x) -> BN(cat_2x2(spikes) W^T) fp32 (B,D,ceil(H/2),ceil(W/2),N); `lin` carries
    the reduction weight as digit planes (`digits`) and the BatchNorm as (alpha, beta).  Returns None when the shape is not covered
    (the caller keeps its gather map + spike_gemm)."""
    B, D, H, W, Cc = spikes.shape
    if getattr(lin, "digits", None) is None:
        return None
    d = MsMergeDesc()
    d.spikes = _ptr(spikes, torch.uint8)
    d.digits, d.cscale = _digits(lin.digits)
    d.alpha, d.beta = _ptr(lin.alpha), _ptr(lin.beta)
    d.B, d.D, d.H, d.W, d.C, d.N = B, D, H, W, Cc, lin.N
    if out is None:
        out = torch.empty((B, D, (H + 1) // 2, (W + 1) // 2, lin.N), dtype=torch.float32, device=spikes.device)
    d.out = _ptr(out, torch.float32)
    _note(flop=2 * out.numel() * 4 * Cc, shape=(out.numel() // lin.N, lin.N, 4 * Cc))
    rc = lib().sdf_ms_patch_merge_fwd(C.byref(d), _stream())
    if rc == E_SHAPE:
        return None
    _check(rc, "sdf_ms_patch_merge_fwd")
    return out


def bn_train_fwd(x2, weight, bias, running_mean, running_var, momentum, eps):
    """sdf_bn_train_fwd on a contiguous (R, C) fp32 buffer: (y, save_mean, save_invstd); running stats updated in place."""
    R, Cc = x2.shape
    y = torch.empty_like(x2)
    mean, invstd = torch.empty(Cc, dtype=torch.float32, device=x2.device), torch.empty(Cc, dtype=torch.float32, device=x2.device)
    nbytes = lib().sdf_bn_train_workspace_bytes(C.c_int64(R), C.c_int(Cc))
    ws = torch.empty((nbytes // 8,), dtype=torch.float64, device=x2.device)
    rc = lib().sdf_bn_train_fwd(C.c_void_p(_ptr(x2, torch.float32)), C.c_void_p(_ptr(weight, torch.float32)),
                                C.c_void_p(_ptr(bias, torch.float32)), C.c_void_p(_ptr(y)), C.c_void_p(_ptr(mean)), C.c_void_p(_ptr(invstd)),
                                C.c_void_p(_ptr(running_mean, torch.float32)), C.c_void_p(_ptr(running_var, torch.float32)),
                                C.c_int64(R), C.c_int(Cc), C.c_float(eps), C.c_float(momentum), C.c_void_p(ws.data_ptr()),
                                C.c_int64(nbytes), _stream())
    _check(rc, "sdf_bn_train_fwd")
    return y, mean, invstd


def bn_train_bwd(x2, grad_y, weight, mean, invstd):
    """sdf_bn_train_bwd: (grad_x, grad_weight, grad_bias)."""
    R, Cc = x2.shape
    gx = torch.empty_like(x2)
    gw, gb = torch.empty(Cc, dtype=torch.float32, device=x2.device), torch.empty(Cc, dtype=torch.float32, device=x2.device)
    nbytes = lib().sdf_bn_train_workspace_bytes(C.c_int64(R), C.c_int(Cc))
    ws = torch.empty((nbytes // 8,), dtype=torch.float64, device=x2.device)
    rc = lib().sdf_bn_train_bwd(C.c_void_p(_ptr(x2, torch.float32)), C.c_void_p(_ptr(grad_y, torch.float32)),
                                C.c_void_p(_ptr(weight, torch.float32)), C.c_void_p(_ptr(mean)), C.c_void_p(_ptr(invstd)),
                                C.c_void_p(_ptr(gx)), C.c_void_p(_ptr(gw)), C.c_void_p(_ptr(gb)), C.c_int64(R), C.c_int(Cc),
                                C.c_void_p(ws.data_ptr()), C.c_int64(nbytes), _stream())
    _check(rc, "sdf_bn_train_bwd")
    return gx, gw, gb


def bn_train_nchw_supported(x):
    return x.dim() == 4 and (x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[1] <= 2048


def bn_train_nchw_fwd(x, weight, bias, running_mean, running_var, momentum, eps):
    """sdf_bn_train_nchw_fwd on a contiguous (N, C, H, W) fp32 buffer: (y, save_mean, save_invstd)."""
    N, Cc, H, W = x.shape
    y = torch.empty_like(x)
    mean, invstd = torch.empty(Cc, dtype=torch.float32, device=x.device), torch.empty(Cc, dtype=torch.float32, device=x.device)
    nbytes = lib().sdf_bn_train_workspace_bytes(C.c_int64(N * H * W), C.c_int((Cc + 3) // 4 * 4))
    ws = torch.empty((nbytes // 8,), dtype=torch.float64, device=x.device)
    rc = lib().sdf_bn_train_nchw_fwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(weight, torch.float32)),
                                     C.c_void_p(_ptr(bias, torch.float32)), C.c_void_p(_ptr(y)), C.c_void_p(_ptr(mean)),
                                     C.c_void_p(_ptr(invstd)), C.c_void_p(_ptr(running_mean, torch.float32)),
                                     C.c_void_p(_ptr(running_var, torch.float32)), C.c_int64(N), C.c_int(Cc), C.c_int(H * W),
                                     C.c_float(eps), C.c_float(momentum), C.c_void_p(ws.data_ptr()), C.c_int64(nbytes), _stream())
    _check(rc, "sdf_bn_train_nchw_fwd")
    return y, mean, invstd


def bn_train_nchw_bwd(x, grad_y, weight, mean, invstd):
    N, Cc, H, W = x.shape
    gx = torch.empty_like(x)
    gw, gb = torch.empty(Cc, dtype=torch.float32, device=x.device), torch.empty(Cc, dtype=torch.float32, device=x.device)
    nbytes = lib().sdf_bn_train_workspace_bytes(C.c_int64(N * H * W), C.c_int((Cc + 3) // 4 * 4))
    ws = torch.empty((nbytes // 8,), dtype=torch.float64, device=x.device)
    rc = lib().sdf_bn_train_nchw_bwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(grad_y, torch.float32)),
                                     C.c_void_p(_ptr(weight, torch.float32)), C.c_void_p(_ptr(mean)), C.c_void_p(_ptr(invstd)),
                                     C.c_void_p(_ptr(gx)), C.c_void_p(_ptr(gw)), C.c_void_p(_ptr(gb)), C.c_int64(N), C.c_int(Cc),
                                     C.c_int(H * W), C.c_void_p(ws.data_ptr()), C.c_int64(nbytes), _stream())
    _check(rc, "sdf_bn_train_nchw_bwd")
    return gx, gw, gb


def rows_gather(x2, row_map):
    """sdf_rows_gather_fwd: (rows, C) fp32 -> (len(map), C), zero rows where the map is negative."""
    out = torch.empty((row_map.numel(), x2.shape[1]), dtype=torch.float32, device=x2.device)
    _check(lib().sdf_rows_gather_fwd(C.c_void_p(_ptr(x2, torch.float32)), C.c_void_p(_ptr(row_map, torch.int32)), C.c_void_p(_ptr(out)),
                                     C.c_int64(row_map.numel()), C.c_int(x2.shape[1]), _stream()), "sdf_rows_gather_fwd")
    return out


def rows_scatter(y2, row_map, rows):
    """sdf_rows_scatter_fwd: (len(map), C) fp32 -> (rows, C): out[map[i]] = y2[i]; rows nobody names are zero."""
    out = torch.zeros((rows, y2.shape[1]), dtype=torch.float32, device=y2.device)
    _check(lib().sdf_rows_scatter_fwd(C.c_void_p(_ptr(y2, torch.float32)), C.c_void_p(_ptr(row_map, torch.int32)), C.c_void_p(_ptr(out)),
                                      C.c_int64(row_map.numel()), C.c_int(y2.shape[1]), _stream()), "sdf_rows_scatter_fwd")
    return out


def qk_gate_f32(q, k, p: NeuronParams):
    """sdf_qk_gate_f32_fwd: e = k * SN2_q(head sums of q) on fp32 spike tensors (Tq, rows, C) (training path)."""
    q, k = q.contiguous(), k.contiguous()
    Tq, Cc = q.shape[0], q.shape[-1]
    e = torch.empty_like(k)
    rc = lib().sdf_qk_gate_f32_fwd(C.c_void_p(_ptr(q, torch.float32)), C.c_void_p(_ptr(k, torch.float32)), C.c_void_p(_ptr(e)),
                                   C.c_int(Tq), C.c_int64(q[0].numel() // Cc), C.c_int(Cc), C.c_int(KIND[p.kind]), C.c_float(p.tau),
                                   C.c_float(p.v_th), C.c_int(1 if p.v_reset is None else 0),
                                   C.c_float(0.0 if p.v_reset is None else p.v_reset), C.c_void_p(_ptr(p.psn_w, torch.float32)),
                                   C.c_void_p(_ptr(p.psn_b, torch.float32)), _stream())
    _check(rc, "sdf_qk_gate_f32_fwd")
    return e


def qk_gate_bwd(q, k, grad_e, p: NeuronParams, detach_reset=True, alpha=2.0):
    """sdf_qk_gate_bwd: (dL/dq, dL/dk, dL/dW, dL/db) of the token gate (the last two only for a PSN gate, else None)."""
    q, k, g = q.contiguous(), k.contiguous(), grad_e.contiguous()
    Tq, Cc = q.shape[0], q.shape[-1]
    rows = q[0].numel() // Cc
    gq, gk = torch.empty_like(q), torch.empty_like(k)
    psn = p.kind == "psn"
    gW = torch.empty((Tq, Tq), dtype=torch.float32, device=q.device) if psn else None
    gb = torch.empty((Tq,), dtype=torch.float32, device=q.device) if psn else None
    nbytes = lib().sdf_qk_gate_bwd_workspace_bytes(C.c_int(Tq), C.c_int64(rows), C.c_int(Cc)) if psn else 0
    ws = torch.empty((max(nbytes, 4) // 4,), dtype=torch.float32, device=q.device) if psn else None
    rc = lib().sdf_qk_gate_bwd(C.c_void_p(_ptr(q, torch.float32)), C.c_void_p(_ptr(k, torch.float32)), C.c_void_p(_ptr(g, torch.float32)),
                               C.c_void_p(_ptr(gq)), C.c_void_p(_ptr(gk)), C.c_int(Tq), C.c_int64(rows), C.c_int(Cc),
                               C.c_int(KIND[p.kind]), C.c_float(p.tau), C.c_float(p.v_th), C.c_int(1 if p.v_reset is None else 0),
                               C.c_float(0.0 if p.v_reset is None else p.v_reset), C.c_int(1 if detach_reset else 0), C.c_int(0),
                               C.c_float(alpha), C.c_void_p(_ptr(p.psn_w, torch.float32)), C.c_void_p(_ptr(p.psn_b, torch.float32)),
                               C.c_void_p(_ptr(gW)), C.c_void_p(_ptr(gb)), C.c_void_p(_ptr(ws)), C.c_int64(nbytes), _stream())
    _check(rc, "sdf_qk_gate_bwd")
    return gq, gk, gW, gb


def affine_resid(x, alpha, beta, Cch, inner, resid=None, out=None):
    """out = fmaf(x, alpha[c], beta[c]) (+ resid) with c = (i // inner) % C (sdf_affine_resid_fwd)."""
    if out is None:
        out = torch.empty_like(x)
    rc = lib().sdf_affine_resid_fwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(alpha, torch.float32)),
                                    C.c_void_p(_ptr(beta, torch.float32)), C.c_void_p(_ptr(resid, torch.float32)),
                                    C.c_void_p(_ptr(out, torch.float32)), C.c_int64(x.numel()), C.c_int(Cch),
                                    C.c_int64(inner), _stream())
    _check(rc, "sdf_affine_resid_fwd")
    return out


def win_attn_ann_windowed(qkv, row_map, B_, N, pad_qkv, scale, bias, mask, nH):
    """The same with the window partition / reverse INSIDE the kernel: qkv (rows, 3C) fp32 in the activation's own row order,
    row_map (B_*N,) int32 from `window_slice_map` (pad + roll + partition; -1 = padding token, which reads `pad_qkv` (3C) and
    writes nothing) -> (rows, C) in the activation's row order (window reverse + roll back + crop), untouched rows zero."""
    rows, C3 = qkv.shape
    out = torch.zeros((rows, C3 // 3), dtype=torch.float32, device=qkv.device)
    d = WinAttnDesc()
    d.mode, d.q, d.k, d.v, d.out = 0, _ptr(qkv, torch.float32), _ptr(qkv), _ptr(qkv), _ptr(out)
    d.B_, d.nW, d.nH, d.N, d.hd = B_, (mask.shape[0] if mask is not None else 1), nH, N, C3 // 3 // nH
    d.scale, d.bias, d.mask = _ptr(scale, torch.float32), _ptr(bias, torch.float32), _ptr(mask, torch.float32)
    d.row_map, d.pad_qkv = _ptr(row_map, torch.int32), _ptr(pad_qkv, torch.float32)
    _check(lib().sdf_win_attn_fwd(C.byref(d), _stream()), "sdf_win_attn_fwd")
    return out


def ann_attn_block_supported(Cc, nH, N):
    """Mirror of sdf_ann_attn_block_supported: the one-launch attention half block is built for C = 96, three heads, windows of 162.
    SDF_ANN_BLOCK=0 (A/B) and SDF_DENSE_LINEAR=0 (library GEMMs for every Linear, the two inside this kernel included) select the
    four-launch path."""
    return (Cc, nH, N) == (96, 3, 162) and sw("SDF_ANN_BLOCK", "1") != "0" and sw("SDF_DENSE_LINEAR", "1") != "0"


def pack_ann_attn_block_weights(wqkv, wproj, nH):
    """fp16 hi / lo planes of the qkv weight (3C, C) as it stands and of the projection weight (C, C) with the 32 input channels of
    every head in the kernel's accumulator order (include/sdformerflow_hip.h: SdfAnnAttnBlockDesc) -> (planes (2,3C,C), planes (2,C,C))."""
    def planes(w):
        w = w.detach().float()
        hi = w.half()
        return torch.stack([hi, (w - hi.float()).half()]).contiguous()
    Cc = wproj.shape[0]
    j = torch.arange(32)
    a, i = j // 8, j % 8
    perm = torch.where(i < 4, 4 * a + i, 16 + 4 * a + i - 4)
    idx = (torch.arange(nH)[:, None] * 32 + perm[None, :]).reshape(-1).to(wproj.device)
    return planes(wqkv), planes(wproj.detach()[:, idx])


def _hi_lo_planes(w):
    w = w.detach().float()
    hi = w.half()
    return torch.stack([hi, (w - hi.float()).half()]).contiguous()


def _acc_order(n32, device):
    """Column permutation of n32 groups of 32 channels into the order the 16 x 16 MFMA accumulators of two neighbouring tiles leave them."""
    j = torch.arange(32)
    a, i = j // 8, j % 8
    perm = torch.where(i < 4, 4 * a + i, 16 + 4 * a + i - 4)
    return (torch.arange(n32)[:, None] * 32 + perm[None, :]).reshape(-1).to(device)


def ann_mlp_block_supported(Cc, Ch):
    """Mirror of sdf_ann_mlp_block_supported: the one-launch MLP half block is built for C = 96, hidden 384.  SDF_ANN_MLP=0 (A/B) and
    SDF_DENSE_LINEAR=0 (library GEMMs for every Linear) select the three-launch path."""
    return (Cc, Ch) == (96, 384) and sw("SDF_ANN_MLP", "1") != "0" and sw("SDF_DENSE_LINEAR", "1") != "0"


def pack_ann_mlp_block_weights(w1, w2):
    """fp16 hi / lo planes of fc1's weight (Ch, C) as it stands and of fc2's weight (C, Ch) with the hidden channels in the kernel's
    accumulator order (include/sdformerflow_hip.h: SdfAnnMlpBlockDesc) -> (planes (2, Ch, C), planes (2, C, Ch))."""
    return _hi_lo_planes(w1), _hi_lo_planes(w2.detach()[:, _acc_order(w2.shape[1] // 32, w2.device)])


def ann_mlp_block(x, out, ln_w, ln_b, ln_eps, w1_planes, b1, w2_planes, b2):
    """sdf_ann_mlp_block_fwd: out = x + fc2(gelu(fc1(LayerNorm(x)))) on rows x (rows, C) fp32, one launch (reference
    swin_transformer3D_v2.py:312-336).  out may be x."""
    rows, Cc = x.shape
    d = AnnMlpBlockDesc()
    d.x, d.out, d.rows, d.C, d.Ch = _ptr(x, torch.float32), _ptr(out, torch.float32), rows, Cc, w1_planes.shape[1]
    d.ln_w, d.ln_b, d.ln_eps = _ptr(ln_w, torch.float32), _ptr(ln_b, torch.float32), float(ln_eps)
    d.w1, d.b1, d.w2, d.b2 = _ptr(w1_planes, torch.float16), _ptr(b1, torch.float32), _ptr(w2_planes, torch.float16), _ptr(b2, torch.float32)
    _check(lib().sdf_ann_mlp_block_fwd(C.byref(d), _stream()), "sdf_ann_mlp_block_fwd")
    return out


LOG2E = 1.4426950408889634


def ann_attn_block_table(scale, bias, mask):
    """(logit scale * log2 e (nH,), ((bias[h] + mask[w]) * log2 e) (nW, nH, N, N)) - what sdf_ann_attn_block_fwd reads: the softmax runs
    in the log2 domain and the additions of the score are the matrix pipe's accumulator input.  Once per parameter version / mask."""
    t = bias[None] if mask is None else bias[None] + mask[:, None]
    return (scale * LOG2E).contiguous(), (t * LOG2E).contiguous()


def ann_attn_block(x, out, row_map, B_, N, nH, ln_w, ln_b, ln_eps, wqkv_planes, qkv_bias, scale2, table, wproj_planes, proj_bias):
    """sdf_ann_attn_block_fwd: out = x + proj(window attention(LayerNorm(x))) on rows x (rows, C) fp32 through the slice map, one launch
    (reference swin_transformer3D_v2.py:272-310 + :331).  (scale2, table) = ann_attn_block_table(...).  out may be x."""
    rows, Cc = x.shape
    d = AnnAttnBlockDesc()
    d.x, d.out, d.row_map = _ptr(x, torch.float32), _ptr(out, torch.float32), _ptr(row_map, torch.int32)
    d.B_, d.nW, d.nH, d.N, d.C, d.rows = B_, table.shape[0], nH, N, Cc, rows
    d.ln_w, d.ln_b, d.ln_eps = _ptr(ln_w, torch.float32), _ptr(ln_b, torch.float32), float(ln_eps)
    d.wqkv, d.qkv_bias = _ptr(wqkv_planes, torch.float16), _ptr(qkv_bias, torch.float32)
    d.scale, d.table = _ptr(scale2, torch.float32), _ptr(table, torch.float32)
    d.wproj, d.proj_bias = _ptr(wproj_planes, torch.float16), _ptr(proj_bias, torch.float32)
    _check(lib().sdf_ann_attn_block_fwd(C.byref(d), _stream()), "sdf_ann_attn_block_fwd")
    return out


def win_attn_ann(qkv, scale, bias, mask, nH):
    """Fused cosine window attention (sdf_win_attn_fwd, SDF_ATTN_ANN): qkv (B_,N,3C) fp32 -> (B_,N,C)."""
    B_, N, C3 = qkv.shape
    out = torch.empty((B_, N, C3 // 3), dtype=torch.float32, device=qkv.device)
    d = WinAttnDesc()
    d.mode, d.q, d.k, d.v, d.out = 0, _ptr(qkv, torch.float32), _ptr(qkv), _ptr(qkv), _ptr(out)
    d.B_, d.nW, d.nH, d.N, d.hd = B_, (mask.shape[0] if mask is not None else 1), nH, N, C3 // 3 // nH
    d.scale, d.bias, d.mask = _ptr(scale, torch.float32), _ptr(bias, torch.float32), _ptr(mask, torch.float32)
    _check(lib().sdf_win_attn_fwd(C.byref(d), _stream()), "sdf_win_attn_fwd")
    return out


def win_attn_sew(q, k, v, scale, bias, mask, nH, Tq, B_, N1):
    """Fused spiking window attention without softmax (SDF_ATTN_SEW): q,k,v u8 (T',B_,N1,C) -> fp32 (T',B_,N1,C)."""
    Cc = q.shape[-1]
    out = torch.empty((Tq, B_, N1, Cc), dtype=torch.float32, device=q.device)
    d = WinAttnDesc()
    d.mode, d.q, d.k, d.v, d.out = 1, _ptr(q, torch.uint8), _ptr(k, torch.uint8), _ptr(v, torch.uint8), _ptr(out)
    d.B_, d.nW, d.nH, d.N, d.hd = B_, (mask.shape[0] if mask is not None else 1), nH, Tq * N1, Cc // nH
    d.Tq, d.N1 = Tq, N1
    d.scale, d.bias, d.mask = _ptr(scale, torch.float32), _ptr(bias, torch.float32), _ptr(mask, torch.float32)
    _check(lib().sdf_win_attn_fwd(C.byref(d), _stream()), "sdf_win_attn_fwd")
    return out


# ---- dense 3x3 convolution of real-valued activations (ANN patch embedding; csrc/dense_conv_wres.hip) ----
def pack_planes(x):
    """(imgs, C, H, W) fp32 -> activation planes (imgs, ceil(C/16), H, W, 32) fp16: per pixel and 16 channels four pieces of
    {4 x hi, 4 x lo}, value = hi + lo (sdf_pack_planes)."""
    x = x.contiguous()
    imgs, Cc, H, W = x.shape
    planes = torch.empty((imgs, -(-Cc // 16), H, W, 32), dtype=torch.float16, device=x.device)
    _check(lib().sdf_pack_planes(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(planes.data_ptr()), C.c_int(imgs), C.c_int(Cc),
                                 C.c_int(H), C.c_int(W), _stream()), "sdf_pack_planes")
    return planes


def unpack_planes(planes, Cc):
    """activation planes -> (imgs, Cc, H, W) fp32 (sdf_unpack_planes)."""
    imgs, nch, H, W, _ = planes.shape
    x = torch.empty((imgs, Cc, H, W), dtype=torch.float32, device=planes.device)
    _check(lib().sdf_unpack_planes(C.c_void_p(_ptr(planes, torch.float16)), C.c_void_p(x.data_ptr()), C.c_int(imgs), C.c_int(Cc),
                                   C.c_int(H), C.c_int(W), _stream()), "sdf_unpack_planes")
    return x


def pack_planes_up2(x, planes=None, rec0=0):
    """Bilinear x2 upsampling (F.interpolate(scale_factor=2, mode="bilinear", align_corners=False)) of x (imgs, C, h, w) fp32 - any
    strides, e.g. a channels-last view - straight into activation planes at twice the size; with `planes` / `rec0` it fills records
    rec0 .. of an existing (imgs, R, 2h, 2w, 32) tensor, which is how a channel concatenation is assembled (sdf_pack_planes_up2)."""
    imgs, Cc, h, w = x.shape
    if planes is None:
        planes = torch.empty((imgs, -(-Cc // 16), 2 * h, 2 * w, 32), dtype=torch.float16, device=x.device)
    if tuple(planes.shape[2:]) != (2 * h, 2 * w, 32) or planes.shape[0] != imgs or not planes.is_contiguous():
        raise SdfError("planes must be a contiguous (imgs, R, 2h, 2w, 32) tensor")
    sn, sc, sh, sw = x.stride()
    _check(lib().sdf_pack_planes_up2(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(planes, torch.float16)), C.c_int(imgs),
                                     C.c_int(Cc), C.c_int(h), C.c_int(w), C.c_int64(sn), C.c_int64(sc), C.c_int64(sh), C.c_int64(sw),
                                     C.c_int(rec0), C.c_int(planes.shape[1]), _stream()), "sdf_pack_planes_up2")
    return planes


def pack_planes_zero_up2(x, planes, rec0=0):
    """Zero insertion x2 of x (imgs, C, h, w) fp32 - any strides - into records rec0 .. of the planes tensor (imgs, R, 2h, 2w, 32): output
    pixel (2k, 2l) = x[k, l], every other pixel 0 (sdf_pack_planes_zero_up2): the input of a stride-2 transposed convolution as a stride-1
    correlation, without the zero-filled fp32 image."""
    imgs, Cc, h, w = x.shape
    if tuple(planes.shape[2:]) != (2 * h, 2 * w, 32) or planes.shape[0] != imgs or not planes.is_contiguous():
        raise SdfError("planes must be a contiguous (imgs, R, 2h, 2w, 32) tensor")
    sn, sc, sh, sw_ = x.stride()
    _check(lib().sdf_pack_planes_zero_up2(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(planes, torch.float16)), C.c_int(imgs),
                                          C.c_int(Cc), C.c_int(h), C.c_int(w), C.c_int64(sn), C.c_int64(sc), C.c_int64(sh), C.c_int64(sw_),
                                          C.c_int(rec0), C.c_int(planes.shape[1]), _stream()), "sdf_pack_planes_zero_up2")
    return planes


def pack_dense_conv_weight(w):
    """Conv2d weight (Cout, Cin, 3, 3) fp32 -> fp16 planes (2, Cout, 9 * 16 * ceil(Cin/16)), w = plane0 + plane1, K ordered
    (16-channel record, ky, kx, channel in record); channels beyond Cin are zero."""
    Cout, Cin, KH, KW = w.shape
    if (KH, KW) != (3, 3):
        raise SdfError("dense_conv3x3 takes 3x3 kernels")
    rec = -(-Cin // 16)
    wp = torch.zeros((Cout, rec * 16, 3, 3), dtype=torch.float32, device=w.device)
    wp[:, :Cin] = w.detach().float()
    wk = wp.view(Cout, rec, 16, 3, 3).permute(0, 1, 3, 4, 2).reshape(Cout, rec * 144)
    return _f16_planes(wk)


def dense_conv_slices(records):
    """Record ranges [(first, count)] a convolution over `records` input records is chained from: sixes, then at most one single."""
    six, one = divmod(records, 6)
    if one > 1:
        return None
    return [(6 * i, 6) for i in range(six)] + ([(6 * six, 1)] if one else [])


def dense_conv3x3_wide(xp, wslices, beta=None, relu=False, out_f32=False):
    """3x3 convolution over any record count `dense_conv_slices` can cut: link k multiplies records [first, first + count) with its own
    weight planes and adds the previous link's result (`resid`); the last link adds beta, applies relu and picks the output format."""
    out = None
    for k, ((r0, n), wp) in enumerate(wslices):
        last = k == len(wslices) - 1
        out = dense_conv3x3(xp[:, r0:r0 + n], wp, None, beta if last else None, out, relu and last, out_f32 and last)
    return out


def dense_conv_applicable(imgs, H, W, Cin, Cout):
    """Shapes sdf_dense_conv3x3_fwd has an instantiation for (Cin <= 16 or Cin == 96, Cout in blocks of 32, 31-bit offsets)."""
    rec = -(-Cin // 16)
    return rec in (1, 6) and Cout % 32 == 0 and imgs * H * W * max(rec * 64, Cout * 4) < 1 << 31


def dense_conv3x3(xp, wplanes, alpha=None, beta=None, resid=None, relu=False, out_f32=False):
    """out = act(alpha * conv3x3(x, w) + beta (+ resid)), stride 1, zero padding 1 (sdf_dense_conv3x3_fwd).  xp / resid / the result
    are activation planes; out_f32 returns (imgs, H, W, N) fp32 (channels last) instead."""
    imgs, rec, H, W, _ = xp.shape
    N = wplanes.shape[1]
    if wplanes.shape[2] != rec * 144:
        raise SdfError(f"weight planes of K = {wplanes.shape[2]} against {rec} activation records")
    # a channel slice xp = full[:, r0:r0 + rec] of a wider contiguous planes tensor is addressed in place
    if xp.stride()[1:] != (H * W * 32, W * 32, 32, 1) or xp.stride(0) % (H * W * 32) or xp.stride(0) < rec * H * W * 32:
        raise SdfError("activation planes must be contiguous or a record slice of a contiguous planes tensor")
    x_records = xp.stride(0) // (H * W * 32)
    if resid is not None and tuple(resid.shape) != (imgs, N // 16, H, W, 32):
        raise SdfError("residual planes must have the output's shape")
    out = (torch.empty((imgs, H, W, N), dtype=torch.float32, device=xp.device) if out_f32 else
           torch.empty((imgs, N // 16, H, W, 32), dtype=torch.float16, device=xp.device))
    d = DenseConvDesc()
    d.x, d.w, d.x_records = _ptr(xp, torch.float16), _ptr(wplanes, torch.float16), x_records
    d.alpha, d.beta = _ptr(alpha, torch.float32), _ptr(beta, torch.float32)
    d.resid, d.out = _ptr(resid, torch.float16), out.data_ptr()
    d.imgs, d.H, d.W, d.cin_records, d.N, d.relu, d.out_f32 = imgs, H, W, rec, N, int(relu), int(out_f32)
    d.acc_scale = _dense_scale(wplanes)
    _check(lib().sdf_dense_conv3x3_fwd(C.byref(d), _stream()), "sdf_dense_conv3x3_fwd")
    return out


# ---- Linear layer on real-valued activations (ANN swin blocks; csrc/dense_linear.hip) ----
def _dense_scale(wplanes):
    """1 / weight scale of dense fp16 planes made by pack_dense_linear_weight / pack_dense_conv_weight.  The scale travels as an attribute of
    the planes tensor: a copy (.to(), .clone(), a slice, a state_dict round trip) drops it, and the kernels would then return results
    wrong by a power of two - so a tensor without it is refused."""
    if not hasattr(wplanes, "sdf_acc_scale"):
        raise SdfError("dense fp16 weight planes without their scale: use the tensor returned by pack_dense_*_weight as is")
    return wplanes.sdf_acc_scale


def _f16_planes(w):
    """fp32 (N, K) -> fp16 planes (2, N, K) of s * w: hi = fp16(s w), lo = fp16(s w - hi), s the power of two that puts max|w| in
    [2^14, 2^15) - every weight within 2^-13 of the largest keeps 22 significant bits (unscaled, a weight of 0.02 has an fp16-subnormal
    lo half and keeps 19).  1 / s travels with the tensor as `sdf_acc_scale`; the kernels multiply the accumulator by it (exact)."""
    import math
    w = w.detach().float()
    mx = float(w.abs().max()) if w.numel() else 0.0
    scale = 2.0 ** (14 - math.floor(math.log2(mx))) if mx > 0 and math.isfinite(mx) else 1.0
    scale = min(max(scale, 2.0 ** -100), 2.0 ** 100)
    ws = w * scale
    hi = ws.half()
    planes = torch.stack([hi, (ws - hi.float()).half()]).contiguous()
    planes.sdf_acc_scale = 1.0 / scale
    return planes


def pack_dense_linear_weight(w):
    """Linear weight (N, K) fp32 -> scaled fp16 planes (2, N, K) (see _f16_planes)."""
    return _f16_planes(w)


def dense_linear_applicable(M, N, K):
    return N % 96 == 0 and K % 32 == 0 and M * max(N, K) * 4 < 1 << 31


def dense_linear(a, wplanes, bias=None, gelu=False, resid=None, out=None):
    """out = act(a @ w.T + bias) (+ resid) for a (M, K) fp32, w as fp16 planes (2, N, K) (sdf_dense_linear_fwd); `out` may be
    `resid` (in-place shortcut add)."""
    M, K = a.shape
    N = wplanes.shape[1]
    if wplanes.shape[2] != K or not a.is_contiguous():
        raise SdfError("dense_linear needs a contiguous (M, K) activation and (2, N, K) weight planes")
    if resid is not None and (tuple(resid.shape) != (M, N) or not resid.is_contiguous()):
        raise SdfError("residual must be a contiguous (M, N) tensor")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    d = DenseLinearDesc()
    d.a, d.w, d.bias, d.resid, d.out = _ptr(a, torch.float32), _ptr(wplanes, torch.float16), _ptr(bias, torch.float32), _ptr(resid, torch.float32), _ptr(out, torch.float32)
    d.M, d.N, d.K, d.gelu = M, N, K, int(gelu)
    d.acc_scale = _dense_scale(wplanes)
    _check(lib().sdf_dense_linear_fwd(C.byref(d), _stream()), "sdf_dense_linear_fwd")
    return out


def linear_train_applicable(M, N, K):
    return N % 96 == 0 and K % 96 == 0 and M * max(N, K) * 4 < 1 << 31


def linear_train(a, w, bias=None, mode=0, conv_wp=0):
    """Training-path Linear products from fp32 tensors (sdf_linear_train_fwd).  mode 0: a (M, K) spikes @ w.T + bias -> (M, N);
    mode 1: a = dY (M, N) @ w -> (M, K).  w is the layer's (N, K) fp32 weight as it is.  conv_wp > 0 (mode 0): the convolution form -
    a (M, C) ringed channels-last rows, w (N, 9 C) ordered (ky, kx, c), result on the ringed grid."""
    N, K = w.shape
    M = a.shape[0]
    if a.shape[1] != ((K // 9 if conv_wp else K) if mode == 0 else N) or not a.is_contiguous() or not w.is_contiguous():
        raise SdfError("linear_train needs contiguous a (M, K | N) and w (N, K)")
    out = torch.empty((M, N if mode == 0 else K), dtype=torch.float32, device=a.device)
    d = LinearTrainDesc()
    d.a, d.w, d.bias, d.out = _ptr(a, torch.float32), _ptr(w, torch.float32), _ptr(bias if mode == 0 else None, torch.float32), _ptr(out, torch.float32)
    d.M, d.N, d.K, d.mode = M, N, K, mode
    d.cv_C, d.cv_Wp = (K // 9, conv_wp) if conv_wp else (0, 0)
    _note(flop=2.0 * M * N * K, bytes=4.0 * M * (N + a.shape[1]), what=f"linear {'fwd' if mode == 0 else 'dX'} {M}x{N}x{K}")
    _check(lib().sdf_linear_train_fwd(C.byref(d), _stream()), "sdf_linear_train_fwd")
    return out


def linear_dw_applicable(M, N, K):
    return N % 96 == 0 and K % 96 == 0 and M * max(N, K) * 4 < 1 << 31


def linear_dw(dy, x, conv_wp=0):
    """dW (N, K) = dy.T @ x for dy (M, N) fp32 and x (M, K) fp32 holding spikes (values exact in bf16) - the weight gradient of a
    spike-fed Linear layer (sdf_linear_dw_fwd): dy split into three bf16 planes in the kernel, fp32 accumulation, m ranges summed in a
    fixed order.  conv_wp > 0: the convolution form on zero-ringed channels-last pixel rows (see conv3x3_dw) -> (N, 9 C)."""
    M, N = dy.shape
    Cx = x.shape[1]
    K = 9 * Cx if conv_wp else Cx
    if x.shape[0] != M or not dy.is_contiguous() or not x.is_contiguous():
        raise SdfError("linear_dw needs contiguous (M, N) and (M, K) operands")
    ns = lib().sdf_linear_dw_splits(C.c_int64(M), C.c_int(N), C.c_int(K), C.c_int(Cx if conv_wp else 0))
    if ns < 1:
        raise SdfError(f"sdf_linear_dw_fwd: unsupported shape M={M} N={N} K={K} (N and K must be multiples of 96)")
    dw = torch.empty((N, K), dtype=torch.float32, device=dy.device)
    part = torch.empty((ns, N, K), dtype=torch.float32, device=dy.device) if ns > 1 else None
    d = LinearDwDesc()
    d.dy, d.x, d.dw, d.partial = _ptr(dy, torch.float32), _ptr(x, torch.float32), _ptr(dw, torch.float32), _ptr(part, torch.float32)
    d.M, d.N, d.K, d.nsplit = M, N, K, ns
    d.cv_C, d.cv_Wp = (Cx, conv_wp) if conv_wp else (0, 0)
    _note(flop=2.0 * M * N * K, bytes=4.0 * M * (N + Cx), what=f"linear dW {M}x{N}x{K}")
    _check(lib().sdf_linear_dw_fwd(C.byref(d), _stream()), "sdf_linear_dw_fwd")
    return dw


def conv3x3_dw_applicable(imgs, Cin, Cout, H, W):
    return Cin % 96 == 0 and Cout % 96 == 0 and imgs * (H + 2) * (W + 2) * max(Cin, Cout) * 4 < 1 << 31


def ringed_rows(t):
    """(imgs, C, H, W) fp32 -> zero-ringed channels-last pixel rows (imgs * (H + 2) * (W + 2), C) (sdf_ringed_rows_fwd)."""
    imgs, Cc, H, W = t.shape
    if not t.is_contiguous():
        raise SdfError("ringed rows need a contiguous (imgs, C, H, W) tensor")
    out = torch.empty((imgs * (H + 2) * (W + 2), Cc), dtype=torch.float32, device=t.device)
    _check(lib().sdf_ringed_rows_fwd(C.c_void_p(_ptr(t, torch.float32)), C.c_void_p(out.data_ptr()), C.c_int(imgs), C.c_int(Cc), C.c_int(H),
                                     C.c_int(W), _stream()), "sdf_ringed_rows_fwd")
    return out


def unring_rows(rows, imgs, Cc, H, W, bias=None):
    """Ringed channels-last rows (imgs * (H + 2) * (W + 2), C) -> (imgs, C, H, W) (+ bias) (sdf_unring_rows_fwd)."""
    if tuple(rows.shape) != (imgs * (H + 2) * (W + 2), Cc) or not rows.is_contiguous():
        raise SdfError("unring_rows: shape does not match the ringed grid")
    out = torch.empty((imgs, Cc, H, W), dtype=torch.float32, device=rows.device)
    _check(lib().sdf_unring_rows_fwd(C.c_void_p(_ptr(rows, torch.float32)), C.c_void_p(_ptr(bias, torch.float32)), C.c_void_p(out.data_ptr()),
                                     C.c_int(imgs), C.c_int(Cc), C.c_int(H), C.c_int(W), _stream()), "sdf_unring_rows_fwd")
    return out


def conv3x3_fwd_ringed(xr, weight, bias, imgs, H, W):
    """Forward of a 3x3 / stride 1 / pad 1 convolution on the ringed rows xr (imgs * (H + 2) * (W + 2), Cin) of a SPIKE image ->
    (imgs, Cout, H, W) fp32 (sdf_linear_train_fwd, convolution form + sdf_unring_rows_fwd)."""
    N = weight.shape[0]
    w9 = weight.detach().float().permute(0, 2, 3, 1).reshape(N, -1).contiguous()
    return unring_rows(linear_train(xr, w9, None, mode=0, conv_wp=W + 2), imgs, N, H, W, bias)


_ringed_rows = ringed_rows


def conv3x3_dw(dy, x):
    """Weight gradient (Cout, Cin, 3, 3) of a 3x3 / stride 1 / pad 1 convolution fed by spikes: dy (imgs, Cout, H, W), x (imgs, Cin,
    H, W) fp32 NCHW as the training path holds them (sdf_linear_dw_fwd, convolution form).  Both go to zero-ringed channels-last
    pixel rows first - on that grid a tap is a row offset, and the zero ring of dy removes what a row offset wraps around an image
    edge."""
    imgs, N, H, W = dy.shape
    Cc = x.shape[1]
    if tuple(x.shape) != (imgs, Cc, H, W) and tuple(x.shape) != (imgs * (H + 2) * (W + 2), Cc):
        raise SdfError("conv3x3_dw needs dy (imgs, Cout, H, W) and x (imgs, Cin, H, W) or its ringed rows")
    xr = x if x.dim() == 2 else ringed_rows(x.contiguous())           # (the forward may have kept the ringed rows)
    dw = linear_dw(ringed_rows(dy), xr, conv_wp=W + 2)
    return dw.view(N, 3, 3, Cc).permute(0, 3, 1, 2).contiguous()


def layer_norm(x, gamma, beta, eps):
    """LayerNorm over the last dim of a contiguous fp32 tensor (sdf_layer_norm_fwd)."""
    Cc = x.shape[-1]
    if not x.is_contiguous():
        raise SdfError("layer_norm needs a contiguous tensor")
    out = torch.empty_like(x)
    _check(lib().sdf_layer_norm_fwd(C.c_void_p(_ptr(x, torch.float32)), C.c_void_p(_ptr(gamma, torch.float32)), C.c_void_p(_ptr(beta, torch.float32)),
                                    C.c_void_p(out.data_ptr()), C.c_int64(x.numel() // Cc), C.c_int(Cc), C.c_float(eps), _stream()),
           "sdf_layer_norm_fwd")
    return out


def dense_conv3x3_strided(x_cl, wplanes, bias, stride, out_T=0):
    """3x3 / pad 1 / stride `stride` convolution of a channels-last fp32 image x_cl (imgs, H, W, C) as a GEMM over gathered rows
    (sdf_dense_linear_fwd, convolution form) -> (imgs, OH, OW, N) fp32 channels last; with out_T = T the images, (t, b)-major on the
    way in, leave (b, t)-major.  wplanes = pack_dense_linear_weight(w.permute(0, 2, 3, 1).reshape(N, 9 * C))."""
    imgs, H, W, Cc = x_cl.shape
    N = wplanes.shape[1]
    if wplanes.shape[2] != 9 * Cc or not x_cl.is_contiguous():
        raise SdfError("dense_conv3x3_strided needs a contiguous (imgs, H, W, C) image and (2, N, 9C) weight planes")
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((imgs, OH, OW, N), dtype=torch.float32, device=x_cl.device)
    d = DenseLinearDesc()
    d.a, d.w, d.bias, d.resid, d.out = _ptr(x_cl, torch.float32), _ptr(wplanes, torch.float16), _ptr(bias, torch.float32), None, _ptr(out)
    d.M, d.N, d.K, d.gelu = imgs * OH * OW, N, 9 * Cc, 0
    d.cv_H, d.cv_W, d.cv_C, d.cv_stride, d.cv_OH, d.cv_OW, d.out_T = H, W, Cc, stride, OH, OW, out_T
    d.acc_scale = _dense_scale(wplanes)
    _check(lib().sdf_dense_linear_fwd(C.byref(d), _stream()), "sdf_dense_linear_fwd")
    return out


def pack_conv_weight(w, nsplit=3, cin_pad=None):
    """Conv2d weight (Cout, Cin, KH, KW) fp32 -> bf16 planes (nsplit, Cout, KH*KW*Cin_pad) in (ky, kx, cin) K order
    (zero rows for padded input channels)."""
    Cout, Cin, KH, KW = w.shape
    cp = Cin if cin_pad is None else cin_pad
    wk = torch.zeros((Cout, KH, KW, cp), dtype=torch.float32, device=w.device)
    wk[..., :Cin] = w.detach().float().permute(0, 2, 3, 1)
    return split_weight(wk.reshape(Cout, KH * KW * cp), nsplit)


def spike_conv2d(x, Wp, imgs, H, W, Cin, OH, OW, KH, KW, stride, dy, dx, out=None, out_spike=None, alpha=None, beta=None,
                 resid=None, out_rowmap=None, sn=None, sn_T=0, pos=None):
    """sdf_spike_conv2d_fwd.  x: u8 spikes NHWC (imgs,H,W,Cin); Wp: planes (nsplit, Cout, KH*KW*Cin).
    fp32 epilogue -> `out` (rows, Cout); fused neuron (sn, sn_T=10, pos=(count, inner, ostride, t_stride)) -> `out_spike`."""
    d = _conv_desc(x, Wp, imgs, H, W, Cin, OH, OW, KH, KW, stride, dy, dx, out, out_spike, alpha, beta, resid, out_rowmap, sn, sn_T, pos)
    g = d.g
    _note(flop=2 * g.M * g.N * g.K, shape=(int(g.M), int(g.N), int(g.K)), fused_neuron=sn is not None)
    _check(lib().sdf_spike_conv2d_fwd(C.byref(d), _stream()), "sdf_spike_conv2d_fwd")
    return out if sn is None else out_spike


def spike_conv2d_multi(x, classes, imgs, H, W, Cin, OH, OW, out, alpha=None, beta=None):
    """sdf_spike_conv2d_multi_fwd: up to four stride-1 convolutions of the same spike images into the same fp32 buffer that differ only in
    taps, weights and output row map - `classes` = [{"Wp", "KH", "KW", "dy", "dx", "rowmap"}, ...], the output-parity classes of a
    transposed convolution - as one launch where the library can (else one call each)."""
    arr = (SpikeConvDesc * len(classes))()
    flop = 0
    for i, c in enumerate(classes):
        d = _conv_desc(x, c["Wp"], imgs, H, W, Cin, OH, OW, c["KH"], c["KW"], 1, c["dy"], c["dx"], out, None, alpha, beta, None, c["rowmap"],
                       None, 0, None)
        C.memmove(C.byref(arr, i * C.sizeof(SpikeConvDesc)), C.byref(d), C.sizeof(SpikeConvDesc))
        flop += 2 * d.g.M * d.g.N * d.g.K
    _note(flop=flop, shape=(int(arr[0].g.M), int(arr[0].g.N), len(classes)))
    _check(lib().sdf_spike_conv2d_multi_fwd(arr, C.c_int(len(classes)), _stream()), "sdf_spike_conv2d_multi_fwd")
    return out


def _conv_desc(x, Wp, imgs, H, W, Cin, OH, OW, KH, KW, stride, dy, dx, out, out_spike, alpha, beta, resid, out_rowmap, sn, sn_T, pos):
    d = SpikeConvDesc()
    g = d.g
    g.A, g.Wp = _ptr(x, torch.uint8), _ptr(Wp)
    g.M, g.N, g.K = imgs * OH * OW, Wp.shape[1], KH * KW * Cin
    g.lda, g.ldo, g.nsplit, g.acc_scale = 0, Wp.shape[1], Wp.shape[0], _acc_scale(Wp)
    if Wp.dtype == torch.int8:                                    # digit planes (split_weight_i8x3)
        g.nsplit, g.col_scale = (PLANES_I8X3_TILED if getattr(Wp, "sdf_tiled", False) else PLANES_I8X3), _ptr(Wp.sdf_col_scale, torch.float32)
    elif Wp.dtype != torch.int16:
        raise SdfError(f"weight planes must be int16 (16-bit float planes) or int8 (digit planes), got {Wp.dtype}")
    g.alpha, g.beta = _ptr(alpha, torch.float32), _ptr(beta, torch.float32)
    g.resid, g.out_rowmap = _ptr(resid, torch.float32), _ptr(out_rowmap, torch.int32)
    if out_rowmap is not None and out is not None:
        g.out_rows = out.numel() // g.ldo
    if sn is not None or Wp.dtype == torch.int8:
        _set_ws(g, x)                                             # (split-K scratch: the small-M digit convolution always splits)
    if sn is not None:
        g.out_spike = _ptr(out_spike, torch.uint8)
        g.out = _ptr(out, torch.float32)                          # optional membrane output (pre-activation + resid)
        g.sn_T, g.sn_kind, g.tau, g.v_th = sn_T, KIND[sn.kind], sn.tau, sn.v_th
        g.v_reset, g.soft_reset = (0.0 if sn.v_reset is None else float(sn.v_reset)), (1 if sn.v_reset is None else 0)
        g.psn_w, g.psn_b = _ptr(sn.psn_w, torch.float32), _ptr(sn.psn_b, torch.float32)
        g.pos_count, g.pos_inner, g.pos_ostride, g.t_stride = pos
    else:
        g.out = _ptr(out, torch.float32)
        _set_ws(g, x)
    d.H, d.W, d.Cin, d.OH, d.OW, d.KH, d.KW, d.sy, d.sx = H, W, Cin, OH, OW, KH, KW, stride, stride
    for i in range(3):
        d.dy[i] = dy[i] if i < len(dy) else 0
        d.dx[i] = dx[i] if i < len(dx) else 0
    return d


def head_conv_sn_supported(T, H, W, Cin, Cout):
    """Shapes sdf_head_conv_sn_fwd is built for (see include/sdformerflow_hip.h)."""
    return (Cin, Cout) in ((2, 32), (2, 48), (2, 64), (4, 48)) and T in (5, 10, 20) and W % 16 == 0


def head_conv_sn(x, w, B, T, H, W, p: NeuronParams, alpha=None, beta=None, voxel_bins=None):
    """sdf_head_conv_sn_fwd: x (B*T,H,W,Cin) fp32 NHWC, w (Cout,Cin,3,3) fp32 -> u8 spikes (B,T,H,W,Cout).
    With `voxel_bins`, x is the event voxel (B, bins, 2, H, W) itself, read in place through strides: input channel ci of time
    step t is polarity ci % 2 of bin (ci // 2) * T + t (reference patch embedding, Spiking_modules.py:1775-1784)."""
    Cout, Cin = w.shape[0], w.shape[1]
    if not x.is_contiguous() or not w.is_contiguous():
        raise SdfError("head_conv_sn needs a contiguous input and OIHW weights")
    out = torch.empty((B, T, H, W, Cout), dtype=torch.uint8, device=x.device)
    d = HeadConvDesc()
    if voxel_bins is not None:
        if tuple(x.shape) != (B, voxel_bins, 2, H, W) or Cin * T > 2 * voxel_bins or Cin > 4:
            raise SdfError(f"voxel {tuple(x.shape)} does not hold {Cin} channels x {T} steps")
        hw = H * W
        d.x_sb, d.x_st, d.x_sy, d.x_sx = voxel_bins * 2 * hw, 2 * hw, W, 1
        for ci in range(Cin):
            d.x_sc[ci] = (ci // 2) * T * 2 * hw + (ci % 2) * hw
    d.x, d.w, d.out = _ptr(x, torch.float32), _ptr(w, torch.float32), _ptr(out, torch.uint8)
    d.alpha, d.beta = _ptr(alpha, torch.float32), _ptr(beta, torch.float32)
    d.B, d.T, d.H, d.W, d.Cin, d.Cout = B, T, H, W, Cin, Cout
    d.sn_kind, d.tau, d.v_th = KIND[p.kind], p.tau, p.v_th
    d.v_reset, d.soft_reset = (0.0 if p.v_reset is None else float(p.v_reset)), (1 if p.v_reset is None else 0)
    d.psn_w, d.psn_b = _ptr(p.psn_w, torch.float32), _ptr(p.psn_b, torch.float32)
    _check(lib().sdf_head_conv_sn_fwd(C.byref(d), _stream()), "sdf_head_conv_sn_fwd")
    return out


def flow_out(pred, H, W, scale_y, scale_x):
    """sdf_flow_out_fwd: pred (B,D,h,w,C) fp32 (last-dim stride 1, rows ldp floats apart) -> (B,C,H,W) = nearest-upsampled time sum."""
    B, D, h, w, Cc = pred.shape
    if pred.stride(4) != 1 or pred.stride(2) != w * pred.stride(3) or pred.stride(1) != h * pred.stride(2) or pred.stride(0) != D * pred.stride(1):
        raise SdfError("flow_out needs a (B,D,h,w,C) view of a dense (rows, ldp) buffer")
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=pred.device)
    _check(lib().sdf_flow_out_fwd(C.c_void_p(_ptr(pred, torch.float32)), C.c_void_p(out.data_ptr()), C.c_int(B), C.c_int(D), C.c_int(h),
                                  C.c_int(w), C.c_int64(pred.stride(3)), C.c_int(Cc), C.c_int(H), C.c_int(W), C.c_float(scale_y),
                                  C.c_float(scale_x), _stream()), "sdf_flow_out_fwd")
    return out


class PointwiseConvDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("imgs", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("Cin", C.c_int32), ("N", C.c_int32), ("stride", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32)]


def pointwise_conv_supported(Cin, N):
    return Cin == 96 and N in (96, 192)


def pointwise_conv_f32(x, w, stride, bias=None):
    """sdf_pointwise_conv_f32_fwd: x (imgs,H,W,Cin) fp32 NHWC, w (N,Cin) fp32 -> (imgs,OH,OW,N) fp32, exact fp32 MFMA."""
    imgs, H, W, Cin = x.shape
    N = w.shape[0]
    if not x.is_contiguous() or not w.is_contiguous() or tuple(w.shape) != (N, Cin):
        raise SdfError("pointwise_conv_f32 needs a contiguous NHWC image and an (N, Cin) weight")
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((imgs, OH, OW, N), dtype=torch.float32, device=x.device)
    d = PointwiseConvDesc()
    d.x, d.w, d.bias, d.out = _ptr(x, torch.float32), _ptr(w, torch.float32), _ptr(bias, torch.float32), out.data_ptr()
    d.imgs, d.H, d.W, d.Cin, d.N, d.stride, d.OH, d.OW = imgs, H, W, Cin, N, stride, OH, OW
    _check(lib().sdf_pointwise_conv_f32_fwd(C.byref(d), _stream()), "sdf_pointwise_conv_f32_fwd")
    return out


class PredHeadDesc(C.Structure):
    _fields_ = [("z", C.c_void_p), ("B", C.c_int32), ("D", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("Cin", C.c_int32),
                ("sn_pred", NeuronCfg), ("wgt", C.c_void_p), ("bias", C.c_void_p), ("pred", C.c_void_p), ("flow", C.c_void_p),
                ("H", C.c_int32), ("W", C.c_int32), ("next_spikes", C.c_void_p), ("next_ld", C.c_int32), ("next_z_off", C.c_int32),
                ("next_pred_off", C.c_int32), ("next_zero_off", C.c_int32), ("next_zero_len", C.c_int32), ("sn_next", NeuronCfg),
                ("keep_spikes", C.c_void_p)]


def pred_head_supported(D, Cin, H, W, h, w, sn, sn_next=None):
    """Mirror of sdf_pred_head_fwd's shape rules (csrc/pred_head.hip): the caller keeps the three-launch form otherwise."""
    if Cin not in (96, 192, 384) or D not in (5, 10, 20) or H % h or W % w:
        return False
    if sn.kind == "psn" and D > 10:
        return False
    return sn_next is None or sn_next.kind == sn.kind or {sn.kind, sn_next.kind} <= {"lif", "if"}


def pred_head(z, wgt, bias, sn, H, W, want_pred=True, nxt=None, keep=False):
    """sdf_pred_head_fwd: z (B,D,h,w,Cin) fp32 channel-last -> (pred (B,D,h,w,4) fp32 or None, flow (B,2,H,W) fp32, spikes of
    SN_pred(z) (B,D,h,w,Cin) u8 if `keep`).  `nxt` = (spike image (B,D,h,w,ld) u8 of the next decoder level, its neuron,
    offset of z's slice, offset of the prediction's 4-wide slice, (offset, length) of the padding channels to zero)."""
    B, D, h, w, Cin = z.shape
    d = PredHeadDesc()
    d.z, d.B, d.D, d.h, d.w, d.Cin = _ptr(z, torch.float32), B, D, h, w, Cin
    _ncfg(d.sn_pred, sn)
    d.wgt, d.bias = _ptr(wgt, torch.float32), _ptr(bias, torch.float32)
    pred = torch.empty((B, D, h, w, 4), dtype=torch.float32, device=z.device) if want_pred else None
    flow = torch.empty((B, 2, H, W), dtype=torch.float32, device=z.device) if H is not None else None
    d.pred, d.flow, d.H, d.W = _ptr(pred), _ptr(flow), H or 0, W or 0
    if nxt is not None:
        img, sn_next, z_off, p_off, (zero_off, zero_len) = nxt
        if img.dtype != torch.uint8 or not img.is_contiguous() or tuple(img.shape[:4]) != (B, D, h, w):
            raise SdfError("the next level's spike image must be a contiguous (B,D,h,w,ld) u8 tensor")
        d.next_spikes, d.next_ld, d.next_z_off, d.next_pred_off = img.data_ptr(), img.shape[4], z_off, p_off
        d.next_zero_off, d.next_zero_len = zero_off, zero_len
        _ncfg(d.sn_next, sn_next)
    sp = torch.empty((B, D, h, w, Cin), dtype=torch.uint8, device=z.device) if keep else None
    d.keep_spikes = _ptr(sp)
    _check(lib().sdf_pred_head_fwd(C.byref(d), _stream()), "sdf_pred_head_fwd")
    return pred, flow, sp


def deconv2x2_applicable(imgs, T, H, W, Cin, Cout, fast_only=False):
    """Mirror of sdf_spike_deconv3x3s2_fwd's shape rules (csrc/spike_gemm.hip).  `fast_only`: only where the halo-tile kernel
    (csrc/spike_deconv_wres.hip: 208 padded input channels, at least 4 096 input pixels) serves it - the row-loop kernel's form of the
    product, which takes every other admitted shape, is slower than the parity-class launches it would replace."""
    if sw("SDF_DECONV_GEMM", "1") == "0" or sw("SDF_RES", "1") == "0":
        return False
    rows = imgs * H * W
    ok = T in (10, 20) and imgs % T == 0 and Cin % 16 == 0 and 16 <= Cin <= 256 and Cout % 8 == 0 and Cout >= 8 and \
        rows * Cin < 1 << 31 and rows * 16 * Cout < 1 << 31
    if fast_only:
        ok = ok and Cin == 208 and rows >= 4096 and sw("SDF_DECONV_WRES", "1") != "0"
    return ok


def pack_deconv2x2_weight(w, cin_pad):
    """ConvTranspose2d(3, stride 2, padding 1, output_padding 1) weight (Cin, Cout, 3, 3) -> int8 digit planes of the (4 Cout, 4 cin_pad)
    matrix of sdf_spike_deconv3x3s2_fwd: row (2 py + px) Cout + co, column q cin_pad + c with q = dh + 2 dw the neighbour (a + dh, b + dw);
    output row 2a + py takes kernel row 1 from input row a when py = 0, kernel rows 2 / 0 from input rows a / a + 1 when py = 1 (columns alike)."""
    Cin, Cout = w.shape[:2]
    w = w.detach().float()
    M = torch.zeros((4 * Cout, 4 * cin_pad), dtype=torch.float32, device=w.device)
    tap = {(0, 0): 1, (1, 0): 2, (1, 1): 0}                      # (output parity, input offset) -> kernel index
    for py in range(2):
        for px in range(2):
            for dh in range(2):
                for dw in range(2):
                    if (py, dh) in tap and (px, dw) in tap:
                        q, cls = dh + 2 * dw, 2 * py + px
                        M[cls * Cout:(cls + 1) * Cout, q * cin_pad:q * cin_pad + Cin] = w[:, :, tap[(py, dh)], tap[(px, dw)]].t()
    return split_weight_i8x3(M)


def spike_deconv3x3s2(s, planes, imgs, T, H, W, Cin, Cout, alpha=None, beta=None, out=None, tiled_bn=False):
    """sdf_spike_deconv3x3s2_fwd: s (imgs, H, W, Cin) u8 -> (imgs, 2H, 2W, Cout) fp32 = BN(ConvTranspose2d(3, 2, 1, 1)(s)), one launch.
    alpha / beta: (Cout), or with `tiled_bn` already repeated to the product's (4 Cout) columns."""
    if out is None:
        out = torch.empty((imgs, 2 * H, 2 * W, Cout), dtype=torch.float32, device=s.device)
    d = SpikeDeconvDesc()
    d.spikes, d.digits, d.cscale = _ptr(s, torch.uint8), _ptr(planes, torch.int8), _ptr(planes.sdf_col_scale, torch.float32)
    if alpha is not None and not tiled_bn:
        alpha, beta = alpha.repeat(4).contiguous(), beta.repeat(4).contiguous()
    d.alpha, d.beta = _ptr(alpha, torch.float32), _ptr(beta, torch.float32)
    d.out, d.imgs, d.T, d.H, d.W, d.Cin, d.Cout = _ptr(out, torch.float32), imgs, T, H, W, Cin, Cout
    _check(lib().sdf_spike_deconv3x3s2_fwd(C.byref(d), _stream()), "sdf_spike_deconv3x3s2_fwd")
    return out


def deconv_col2im(Y, imgs, H, W, Cout, alpha=None, beta=None, out=None):
    """sdf_deconv_col2im_fwd: Y (imgs*H*W, 9*Cout) fp32 per-tap products -> (imgs, 2H, 2W, Cout) fp32, BN fused."""
    if out is None:
        out = torch.empty((imgs, 2 * H, 2 * W, Cout), dtype=torch.float32, device=Y.device)
    _check(lib().sdf_deconv_col2im_fwd(C.c_void_p(_ptr(Y, torch.float32)), C.c_void_p(_ptr(alpha, torch.float32)),
                                       C.c_void_p(_ptr(beta, torch.float32)), C.c_void_p(_ptr(out, torch.float32)), C.c_int(imgs),
                                       C.c_int(H), C.c_int(W), C.c_int(Cout), _stream()), "sdf_deconv_col2im_fwd")
    return out
