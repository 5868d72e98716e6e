"""CPU-side checks of the C ABI: the library loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sdformerflow_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|int64_t) (sdf_\w+)\(", src, flags=re.M)))


def test_header_declares_the_expected_entry_points():
    from sdformerflow_amd import hip
    assert set(declared_symbols()) == set(hip.EXPORTS)


def test_library_exports_every_declared_symbol():
    from sdformerflow_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.sdf_version.restype = ctypes.c_int
    assert lib.sdf_version() == 100            # host-only call, no GPU needed


def test_product_refuses_cpu_tensors():
    import torch
    from sdformerflow_amd import hip
    with pytest.raises(hip.SdfError):
        hip.lif_fwd(torch.zeros(2, 8))


def test_argument_errors_are_reported_before_any_launch():
    """Error behaviour of the C ABI (include/sdformerflow_hip.h: negative = argument error, checked before launch - so
    these calls never touch the dummy pointers and need no GPU): NULL, shape, selector and alignment codes."""
    import ctypes as C
    from sdformerflow_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(hip.LIB_PATH)
    for name in hip.EXPORTS[1:]:
        getattr(lib, name).restype = C.c_int
    p, odd = C.c_void_p(0x10000), C.c_void_p(0x10004)
    E_NULL, E_SHAPE, E_DTYPE, E_ALIGN = -1, -2, -3, -4
    lif = lambda x, out, T, N, dt=0: lib.sdf_lif_fwd(x, out, None, C.c_int(T), C.c_int64(N), C.c_float(2.0), C.c_float(0.1),
                                                       C.c_int(1), C.c_float(0.0), C.c_int(dt), None)
    assert lif(None, p, 10, 64) == E_NULL
    assert lif(p, p, 10, 0) == E_SHAPE                          # N < 1 (N % 4 != 0 is served by the library since round 2)
    assert lif(p, p, 0, 64) == E_SHAPE                          # T < 1
    assert lif(p, p, 10, 64, dt=7) == E_DTYPE                   # unknown spike dtype
    assert lif(odd, p, 10, 64) == E_ALIGN
    bwd = lambda x, g, gx, T, N, kind=0, tau=2.0, sur=0: lib.sdf_lif_bwd(
        x, g, gx, C.c_int(T), C.c_int64(N), C.c_int(kind), C.c_float(tau), C.c_float(0.1), C.c_int(1), C.c_float(0.0), C.c_int(1),
        C.c_int(sur), C.c_float(2.0), None)
    assert bwd(p, None, p, 10, 64) == E_NULL
    assert bwd(p, p, p, 10, 66) == E_SHAPE
    assert bwd(p, p, p, 3, 64) == E_SHAPE                       # T outside the compiled set
    assert bwd(p, p, p, 10, 64, tau=1.0) == E_SHAPE             # LIF needs tau > 1
    assert bwd(p, p, p, 10, 64, sur=1) == E_DTYPE               # only the ATan surrogate
    assert bwd(p, p, p, 10, 64, kind=1) == E_DTYPE              # PSN has its own entry point
    assert bwd(p, odd, p, 10, 64) == E_ALIGN
    psn = lambda gW, gb, ws, T: lib.sdf_psn_bwd(p, p, p, p, p, gW, gb, None, ws, C.c_int64(1 << 20), C.c_int(T), C.c_int64(64),
                                                 C.c_int(0), C.c_float(2.0), None)
    assert psn(p, None, p, 10) == E_NULL                        # dW without db
    assert psn(p, p, p, 20) == E_SHAPE                          # in-kernel dW / db reduction only for T <= 10
    lib.sdf_psn_bwd_workspace_bytes.restype = C.c_int64
    assert lib.sdf_psn_bwd_workspace_bytes(C.c_int(10), C.c_int64(1 << 20)) == 512 * 110 * 4
    assert lib.sdf_psn_bwd_workspace_bytes(C.c_int(20), C.c_int64(1 << 20)) == 0
    d = hip.WinAttnDesc()
    d.mode, d.q, d.k, d.v, d.out, d.scale, d.bias = 0, 0x10000, 0x10000, 0x10000, 0x10000, 0x10000, 0x10000
    d.B_, d.nW, d.nH, d.N, d.hd = 4, 1, 3, 162, 16
    assert lib.sdf_win_attn_fwd(C.byref(d), None) == E_SHAPE    # head dim must be 32
    d.hd, d.N = 32, 200
    assert lib.sdf_win_attn_fwd(C.byref(d), None) == E_SHAPE    # at most 192 tokens per window
    d.N, d.mode = 162, 5
    assert lib.sdf_win_attn_fwd(C.byref(d), None) == E_DTYPE
    d.mode, d.bias = 0, None
    assert lib.sdf_win_attn_fwd(C.byref(d), None) == E_NULL
    assert lib.sdf_spike_gemm_fwd(None, None) == E_NULL and lib.sdf_neuron_fwd(None, None) == E_NULL
    assert lib.sdf_qk_attn_fwd(None, None) == E_NULL
    q = hip.QkAttnDesc()
    q.x, q.slice_map, q.workspace, q.p_planes, q.qk_planes = 0x10000, 0x10000, 0x10000, 0x10000, 0x10000
    q.B_, q.Tq, q.N1, q.C, q.nH, q.workspace_bytes = 4, 2, 81, 100, 3, 1 << 30
    assert lib.sdf_qk_attn_fwd(C.byref(q), None) == E_SHAPE     # C must be nH * 32
    q.C, q.workspace_bytes = 96, 16
    assert lib.sdf_qk_attn_fwd(C.byref(q), None) == E_SHAPE     # workspace smaller than sdf_qk_attn_workspace_bytes
    lib.sdf_qk_attn_workspace_bytes.restype = C.c_int64
    assert lib.sdf_qk_attn_workspace_bytes(C.c_int64(4), C.c_int(2), C.c_int(81), C.c_int(96)) == 62208 + 2 * 62208
    assert lib.sdf_window_slice_map(None, 1, 2, 9, 9, 2, 9, 9, 0, 0, 0, None, None) == E_NULL
    assert lib.sdf_window_slice_map(p, 1, 2, 9, 9, 0, 9, 9, 0, 0, 0, None, None) == E_SHAPE


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under sdformerflow_amd/ (nor bench.py outside its cpu_baseline leg) may import
    it, and there is no CPU fallback module to import instead."""
    import ast
    pkg = os.path.join(ROOT, "sdformerflow_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith(".py"):
                continue
            tree = ast.parse(open(os.path.join(dirpath, f)).read())
            for node in ast.walk(tree):
                names = []
                if isinstance(node, ast.Import):
                    names = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    names = [node.module or ""]
                assert not any(n == "oracle" or n.startswith("oracle.") for n in names), (f, names)
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.FunctionDef):
            uses = any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n) for n in ast.walk(node))
            assert not uses or node.name == "cpu_baseline", node.name
