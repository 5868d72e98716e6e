"""CPU-side checks of the C ABI: the library loads and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "sdformerflow_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|int64_t|void) (sdf_\w+)\(", src, flags=re.M)))


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
    assert lib.sdf_version() == 107            # host-only call, no GPU needed


def test_switches_are_read_once_and_reloaded_on_request(monkeypatch):
    """The diagnostic SDF_* switches are a read-once table on both sides of the C ABI (csrc/switches.hip, hip.sw): a change of the
    environment is seen after sdf_switches_reload() / hip.reload_switches() and not before; no entry point calls getenv per call."""
    import subprocess
    from sdformerflow_amd import hip
    hip.lib()
    os.environ["SDF_WIDE_MAXROWS"] = "12345"                        # (behind monkeypatch's back: no reload)
    try:
        assert hip.sw("SDF_WIDE_MAXROWS") is None
        hip.reload_switches()
        assert hip.sw("SDF_WIDE_MAXROWS") == "12345"
    finally:
        os.environ.pop("SDF_WIDE_MAXROWS", None)
        hip.reload_switches()
    assert hip.sw("SDF_WIDE_MAXROWS") is None
    with hip.scoped_switches(SDF_RES="0"):
        assert hip.sw("SDF_RES") == "0"
    assert hip.sw("SDF_RES") is None
    # getenv is referenced by exactly one translation unit of the library: the table builder
    src = os.path.join(ROOT, "sdformerflow_amd", "csrc")
    code = lambda f: "\n".join(l.split("//")[0] for l in open(os.path.join(src, f)).read().split("\n"))      # (comments dropped)
    users = [f for f in sorted(os.listdir(src)) if f.endswith((".hip", ".h")) and "getenv(" in code(f)]
    assert users == ["switches.hip"], users


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
        getattr(lib, name).restype = None if name in hip.VOID_EXPORTS else C.c_int
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
    # the transposed convolution as one product (round 5): T in {10, 20} dividing the images, Cin % 16, 4 Cin <= 1024, Cout % 8
    dc = hip.SpikeDeconvDesc()
    assert lib.sdf_spike_deconv3x3s2_fwd(None, None) == E_NULL and lib.sdf_spike_deconv3x3s2_fwd(C.byref(dc), None) == E_NULL
    dc.spikes = dc.digits = dc.cscale = dc.out = 0x10000
    dc.imgs, dc.T, dc.H, dc.W, dc.Cin, dc.Cout = 10, 10, 72, 96, 208, 48
    dc.alpha = 0x10000
    assert lib.sdf_spike_deconv3x3s2_fwd(C.byref(dc), None) == E_NULL          # alpha without beta
    dc.alpha = None
    for field, bad in (("T", 7), ("imgs", 15), ("Cin", 200), ("Cin", 272), ("Cout", 44)):
        keep = getattr(dc, field)
        setattr(dc, field, bad)
        assert lib.sdf_spike_deconv3x3s2_fwd(C.byref(dc), None) == E_SHAPE, field
        setattr(dc, field, keep)
    # the one-launch MLP half block (round 5): C = 96, hidden 384
    assert lib.sdf_ann_mlp_block_supported(96, 384) == 1 and lib.sdf_ann_mlp_block_supported(192, 768) == 0
    mb = hip.AnnMlpBlockDesc()
    assert lib.sdf_ann_mlp_block_fwd(None, None) == E_NULL and lib.sdf_ann_mlp_block_fwd(C.byref(mb), None) == E_NULL
    for f in ("x", "out", "ln_w", "ln_b", "w1", "w2"):
        setattr(mb, f, 0x10000)
    mb.rows, mb.C, mb.Ch = 1000, 192, 768
    assert lib.sdf_ann_mlp_block_fwd(C.byref(mb), None) == E_SHAPE
    # the one-launch attention half block (round 5): built for C = 96 / 3 heads / 162-token windows, refused before any launch otherwise
    assert lib.sdf_ann_attn_block_supported(96, 3, 162) == 1 and lib.sdf_ann_attn_block_supported(192, 6, 162) == 0
    ab = hip.AnnAttnBlockDesc()
    assert lib.sdf_ann_attn_block_fwd(None, None) == E_NULL and lib.sdf_ann_attn_block_fwd(C.byref(ab), None) == E_NULL
    for f in ("x", "out", "row_map", "ln_w", "ln_b", "wqkv", "wproj", "scale", "table"):
        setattr(ab, f, 0x10000)
    ab.B_, ab.nW, ab.nH, ab.N, ab.C, ab.rows = 8, 4, 6, 162, 192, 1296
    assert lib.sdf_ann_attn_block_fwd(C.byref(ab), None) == E_SHAPE
    ab.nH, ab.C, ab.nW = 3, 96, 3
    assert lib.sdf_ann_attn_block_fwd(C.byref(ab), None) == E_SHAPE     # windows must be a multiple of the mask tables
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
    # E (padded to 256 bytes), q | k (padded), and the slice spikes of the wide-stage form (round 4)
    assert lib.sdf_qk_attn_workspace_bytes(C.c_int64(4), C.c_int(2), C.c_int(81), C.c_int(96)) == 62208 + 2 * 62208 + 62208
    # wide-stage additions (round 4): the inverse slice map, and the host-only "will this run wide" queries
    assert lib.sdf_window_zsrc_map(None, C.c_int64(4), 2, 81, 12, p, None) == E_NULL
    assert lib.sdf_window_zsrc_map(p, C.c_int64(0), 2, 81, 12, p, None) == E_SHAPE
    assert lib.sdf_qk_attn_is_wide(None) == 0 and lib.sdf_ms_mlp_is_wide(None) == 0
    assert lib.sdf_qk_attn_is_wide(C.byref(q)) == 0              # no inverse map / digit planes given: the general kernels
    q.emit_s1 = 0x10000
    q.workspace_bytes = 1 << 30
    assert lib.sdf_qk_attn_fwd(C.byref(q), None) == E_SHAPE     # only the wide-stage projection emits the next neuron's spikes
    q.emit_s1 = None
    m = hip.MsMlpDesc()
    m.x, m.fc1_planes, m.fc2_planes, m.workspace, m.fc1_alpha, m.fc1_beta, m.fc2_alpha, m.fc2_beta = (0x10000,) * 8
    m.B, m.D, m.HW, m.C, m.Ch, m.nsplit, m.workspace_bytes = 1, 10, 432, 384, 1536, 2, 1 << 30
    m.sn1.kind = m.sn2.kind = hip.SDF_LIF
    m.sn1.tau = m.sn2.tau = 2.0
    assert lib.sdf_ms_mlp_is_wide(C.byref(m)) == 0              # no int8 digit planes given: the general kernels
    m.fc1_digits, m.fc1_cscale, m.fc2_digits, m.fc2_cscale = (0x10000,) * 4
    assert lib.sdf_ms_mlp_is_wide(C.byref(m)) == 1
    m.flags = hip.MLP_NARROW
    assert lib.sdf_ms_mlp_is_wide(C.byref(m)) == 0
    m.flags, m.nsplit = 0, 3
    assert lib.sdf_ms_mlp_is_wide(C.byref(m)) == 0              # the exact three-plane mode keeps the general kernels
    m.nsplit, m.C, m.Ch = 2, 192, 768
    assert lib.sdf_ms_mlp_is_wide(C.byref(m)) == 1              # round 5: stage 1 (C = 192) runs on digit planes too (csrc/ms_res.hip)
    m.C, m.Ch = 96, 384
    assert lib.sdf_ms_mlp_is_wide(C.byref(m)) == 0              # stage 0 keeps the general kernels by default (SDF_RES_MINC = 128)
    m.C, m.Ch, m.s1_in = 224, 896, 0x10000                      # (neither a narrow nor a wide stage: C > 192, C % 64 != 0)
    assert lib.sdf_ms_mlp_is_wide(C.byref(m)) == 0
    assert lib.sdf_ms_mlp_fwd(C.byref(m), None) == E_SHAPE      # s1_in is a digit-plane-stage input
    # round 4, late: the small-M kernel's weight format, patch merging on emitted spikes, several convolutions in one launch
    q16 = C.c_void_p(0x10000)
    assert lib.sdf_tile_weight_i8x3(None, q16, 64, 128, None) == E_NULL
    assert lib.sdf_tile_weight_i8x3(q16, q16, 60, 128, None) == E_SHAPE       # N in blocks of 16
    assert lib.sdf_tile_weight_i8x3(q16, q16, 64, 100, None) == E_SHAPE       # K in steps of 64
    assert lib.sdf_ms_patch_merge_fwd(None, None) == E_NULL
    mg = hip.MsMergeDesc()
    mg.spikes = mg.digits = mg.cscale = mg.out = 0x10000
    mg.B, mg.D, mg.H, mg.W, mg.C, mg.N = 1, 4, 8, 8, 128, 256
    assert lib.sdf_ms_patch_merge_fwd(C.byref(mg), None) == E_SHAPE          # D in {10, 20}
    mg.D, mg.C = 10, 80
    assert lib.sdf_ms_patch_merge_fwd(C.byref(mg), None) == E_SHAPE          # C in steps of 32 up to 192, of 64 beyond
    assert lib.sdf_spike_conv2d_multi_fwd(None, 4, None) == E_NULL
    assert lib.sdf_spike_conv2d_multi_fwd(q16, 0, None) == E_SHAPE
    g5 = hip.SpikeGemmDesc()
    g5.A, g5.Wp, g5.out, g5.col_scale = 0x10000, 0x10000, 0x10000, 0x10000
    g5.M, g5.N, g5.K, g5.lda, g5.ldo, g5.nsplit = 12, 64, 128, 128, 64, hip.PLANES_I8X3_TILED
    assert lib.sdf_spike_gemm_fwd(C.byref(g5), None) == E_SHAPE              # tiled digits: M in multiples of 10 (no other kernel reads them)
    assert lib.sdf_window_slice_map(None, 1, 2, 9, 9, 2, 9, 9, 0, 0, 0, None, None) == E_NULL
    assert lib.sdf_window_slice_map(p, 1, 2, 9, 9, 0, 9, 9, 0, 0, 0, None, None) == E_SHAPE
    # dense kernels of the ANN path (round 2)
    assert lib.sdf_dense_conv3x3_fwd(None, None) == E_NULL and lib.sdf_dense_linear_fwd(None, None) == E_NULL
    dc = hip.DenseConvDesc()
    dc.x, dc.w, dc.out = 0x10000, 0x10000, 0x10000
    dc.imgs, dc.H, dc.W, dc.cin_records, dc.N = 2, 16, 16, 6, 48
    assert lib.sdf_dense_conv3x3_fwd(C.byref(dc), None) == E_SHAPE        # output columns come in blocks of 32
    dc.N, dc.cin_records = 96, 2
    assert lib.sdf_dense_conv3x3_fwd(C.byref(dc), None) == E_SHAPE        # instantiated for 1 and 6 input records
    dc.cin_records, dc.x_records = 6, 3
    assert lib.sdf_dense_conv3x3_fwd(C.byref(dc), None) == E_SHAPE        # a slice cannot be wider than the tensor it is cut from
    dc.x_records, dc.imgs, dc.H, dc.W = 0, 64, 288, 384
    assert lib.sdf_dense_conv3x3_fwd(C.byref(dc), None) == E_SHAPE        # 31-bit offsets
    dc.imgs, dc.x = 2, 0x10004
    assert lib.sdf_dense_conv3x3_fwd(C.byref(dc), None) == E_ALIGN
    dl = hip.DenseLinearDesc()
    dl.a, dl.w, dl.out, dl.M, dl.N, dl.K = 0x10000, 0x10000, 0x10000, 100, 64, 96
    assert lib.sdf_dense_linear_fwd(C.byref(dl), None) == E_SHAPE         # N % 96
    dl.N, dl.K = 96, 48
    assert lib.sdf_dense_linear_fwd(C.byref(dl), None) == E_SHAPE         # K % 32
    dl.K, dl.cv_C, dl.cv_H, dl.cv_W, dl.cv_stride, dl.cv_OH, dl.cv_OW = 9 * 32, 32, 8, 8, 4, 2, 3
    assert lib.sdf_dense_linear_fwd(C.byref(dl), None) == E_SHAPE         # OW must be (W - 1) / stride + 1
    assert lib.sdf_pack_planes(None, p, 1, 16, 4, 4, None) == E_NULL and lib.sdf_pack_planes(p, p, 1, 0, 4, 4, None) == E_SHAPE
    assert lib.sdf_pack_planes_up2(p, p, 1, 16, 4, 4, C.c_int64(256), C.c_int64(16), C.c_int64(4), C.c_int64(1), 1, 1, None) == E_SHAPE   # records past the tensor
    assert lib.sdf_layer_norm_fwd(p, p, p, p, C.c_int64(8), 98, C.c_float(1e-5), None) == E_SHAPE      # C % 4
    assert lib.sdf_layer_norm_fwd(p, p, p, odd, C.c_int64(8), 96, C.c_float(1e-5), None) == E_ALIGN

    # training path: weight gradients (round 5)
    assert lib.sdf_linear_dw_fwd(None, None) == E_NULL
    assert lib.sdf_linear_dw_splits(C.c_int64(276480), 96, 96, 0) == 768 and lib.sdf_linear_dw_splits(C.c_int64(4320), 768, 3072, 0) == 3
    assert lib.sdf_linear_dw_splits(C.c_int64(100), 96, 80, 0) == 0 and lib.sdf_linear_dw_splits(C.c_int64(100), 96, 864, 48) == 0
    assert lib.sdf_linear_dw_splits(C.c_int64(1132960), 96, 864, 96) == 256       # convolution form: three taps per column tile
    dwd = hip.LinearDwDesc()
    dwd.dy, dwd.x, dwd.dw, dwd.M, dwd.N, dwd.K, dwd.nsplit = 0x10000, 0x10000, 0x10000, 100, 96, 80, 1
    assert lib.sdf_linear_dw_fwd(C.byref(dwd), None) == E_SHAPE           # K % 96
    dwd.K, dwd.nsplit = 96, 2
    assert lib.sdf_linear_dw_fwd(C.byref(dwd), None) == E_NULL            # several ranges need the partial buffer
    dwd.nsplit, dwd.cv_C, dwd.cv_Wp = 1, 96, 10
    assert lib.sdf_linear_dw_fwd(C.byref(dwd), None) == E_SHAPE           # convolution form: K = 9 C
    dwd.cv_C, dwd.cv_Wp, dwd.M = 0, 0, 1 << 23
    assert lib.sdf_linear_dw_fwd(C.byref(dwd), None) == E_SHAPE           # 31-bit offsets
    dwd.M, dwd.dy = 100, 0x10004
    assert lib.sdf_linear_dw_fwd(C.byref(dwd), None) == E_ALIGN
    assert lib.sdf_linear_train_fwd(None, None) == E_NULL
    lt = hip.LinearTrainDesc()
    lt.a, lt.w, lt.out, lt.M, lt.N, lt.K, lt.mode = 0x10000, 0x10000, 0x10000, 100, 96, 80, 0
    assert lib.sdf_linear_train_fwd(C.byref(lt), None) == E_SHAPE          # forward: K % 32
    lt.K, lt.mode = 96, 2
    assert lib.sdf_linear_train_fwd(C.byref(lt), None) == E_SHAPE          # mode 0 | 1
    lt.mode, lt.N, lt.K = 1, 96, 64
    assert lib.sdf_linear_train_fwd(C.byref(lt), None) == E_SHAPE          # dX: K % 96 (its output columns)
    lt.K, lt.w = 96, 0x10008
    assert lib.sdf_linear_train_fwd(C.byref(lt), None) == E_ALIGN
    lt.w, lt.mode, lt.N, lt.K, lt.cv_C, lt.cv_Wp = 0x10000, 0, 96, 864, 64, 10
    assert lib.sdf_linear_train_fwd(C.byref(lt), None) == E_SHAPE          # convolution form: K = 9 cv_C
    lt.cv_C, lt.mode = 96, 1
    assert lib.sdf_linear_train_fwd(C.byref(lt), None) == E_SHAPE          # convolution form: forward only
    assert lib.sdf_unring_rows_fwd(None, None, p, 1, 96, 4, 4, None) == E_NULL and lib.sdf_unring_rows_fwd(p, None, p, 1, 100, 4, 4, None) == E_SHAPE
    assert lib.sdf_ringed_rows_fwd(None, p, 1, 96, 4, 4, None) == E_NULL and lib.sdf_ringed_rows_fwd(p, p, 1, 48, 4, 4, None) == E_SHAPE


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
