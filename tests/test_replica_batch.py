"""Replicas: R independent batch-1 forwards served by ONE launch sequence (`model.forward_replicas`, engine.forward(replicas=True)).

The reference couples the samples of a batch (`window_partition_v2`'s raw `.view(Wd, B_, ...)` and the attention's raw head
reshape, Spiking_swin_transformer3D.py:100-113, :709-710), so "a batch of R" is NOT R forwards.  Replicas keep every
sample's batch-1 view and only concatenate the index tables; CPU tests restate both tables against the reference's batch-1
expressions, the GPU tests hold the flows to bit-equality with separate batch-1 forwards."""
import numpy as np
import pytest
import torch

from sdformerflow_amd import hip
from sdformerflow_amd.STSwinNet_SNN.Spiking_swin_transformer3D import window_slice_map


def zsrc_batch1(m1, nW, Tq, N1, nH, x_rows):
    """numpy restatement of sdf_window_zsrc_map (csrc/ms_wide.hip zsrc_kernel): per row of x the offset of (head 0, byte 0) of its
    gated spikes in E_flat behind Z[t,b,n,g,d] = E_flat[((((b nH + g) T' + t) N1 + n) 32 + d]."""
    z = np.zeros(x_rows, np.int64)
    i = np.arange(Tq * nW * N1)
    n, sl = i % N1, i // N1
    t, b = sl // nW, sl % nW
    ok = m1 >= 0
    z[m1[ok]] = ((((b * nH) * Tq + t) * N1 + n) * 32)[ok]
    return z


@pytest.mark.parametrize("D,H,W,ws,ss,nH,R", [(10, 18, 24, (2, 9, 9), (0, 0, 0), 3, 3), (10, 18, 24, (2, 9, 9), (1, 4, 4), 3, 2),
                                                (4, 9, 12, (2, 9, 9), (1, 4, 4), 6, 4), (6, 20, 17, (2, 5, 5), (1, 2, 2), 2, 3)])
def test_replica_tables_are_the_batch1_tables_per_sample(D, H, W, ws, ss, nH, R):
    """Gathering through the replica slice map = every sample gathered through its own batch-1 map, laid out (T', R nW, N1); reading E
    through the replica operand map + the constant head stride = every sample's own raw head reshape of ITS (T', nW, N1, C) tensor."""
    Tq, N1, Cc = ws[0], ws[1] * ws[2], nH * 32
    m1, nW = window_slice_map(1, D, H, W, ws, ss)
    m1 = m1.reshape(-1)
    rows1 = D * H * W
    mR = hip.replica_slice_map(torch.from_numpy(m1), nW, R, Tq, N1, rows1).numpy()
    assert mR.shape == (Tq * R * nW * N1,)
    rng = np.random.default_rng(5)
    x = rng.integers(0, 255, size=(R, rows1 + 1), dtype=np.int64)      # one value per row of x (+ a zero row for the padding index -1)
    x[:, -1] = 0
    xflat = np.concatenate([x[:, :-1].reshape(-1), [0]])
    got = xflat[mR].reshape(Tq, R, nW * N1)
    for r in range(R):
        assert np.array_equal(got[:, r], x[r][m1].reshape(Tq, nW * N1))
    # the reference's own batch view is a DIFFERENT table whenever R > 1 (that is the coupling): replicas are not "batch = R"
    mB, _ = window_slice_map(R, D, H, W, ws, ss)
    assert not np.array_equal(mB.reshape(-1), mR)
    # the projection's operand: E is (T', R nW, N1, C) bytes in memory; Z row (t, b, n) of sample r, head g, reads 32 bytes
    z1 = zsrc_batch1(m1, nW, Tq, N1, nH, rows1)
    zR = hip.replica_zsrc_map(torch.from_numpy(z1.astype(np.int32)), nW, R, Tq, N1, Cc).numpy().astype(np.int64)
    E = rng.integers(0, 2, size=(Tq, R, nW, N1, Cc), dtype=np.uint8)
    Eflat = E.reshape(-1)
    G = Tq * N1 * 32                                                      # the kernels' constant head-to-head stride (wide_common.h zg_G)
    for r in range(R):
        Er = np.ascontiguousarray(E[:, r]).reshape(-1)                   # the sample's own E_flat
        Zr = Er.reshape(nW, nH, Tq, N1, 32).transpose(2, 0, 3, 1, 4).reshape(Tq * nW * N1, Cc)      # reference :709-710
        for row in rng.choice(Tq * nW * N1, size=64, replace=False):
            xr = m1[row]                                                  # Z row `row` is scattered to x row m1[row] (the same table)
            if xr < 0:
                continue
            for g in range(nH):
                o = zR[r * rows1 + xr] + g * G
                assert np.array_equal(Eflat[o:o + 32], Zr[row, 32 * g:32 * g + 32])


def test_replica_tables_refuse_a_window_count_the_head_stride_cannot_serve():
    with pytest.raises(hip.ReplicaGeometryError):
        hip.replica_zsrc_map(torch.zeros(9 * 9 * 3, dtype=torch.int32), 3, 2, 2, 81, 96)


DEV = "cuda:0"


def _model(kind, H, W, en4, T=10):
    import yaml, os
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet, MS_SpikingformerFlowNet_en4
    from sdformerflow_amd.synthetic import synth_state_dict
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind, num_steps=T)
    cfg["model"]["num_bins"] = T
    cfg["swin_transformer"]["input_size"] = [H, W]
    cls = MS_SpikingformerFlowNet_en4
    if not en4:
        cls = MS_SpikingformerFlowNet
        cfg["swin_transformer"].update(swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    model = cls(cfg["model"].copy(), cfg["swin_transformer"].copy())
    model.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}), strict=True)
    return model.eval().to(DEV)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,H,W,en4,R", [("lif", 144, 192, False, 3), ("psn", 144, 192, False, 2), ("lif", 288, 384, True, 4),
                                             ("psn", 288, 384, True, 2), ("lif", 288, 384, True, 3), ("lif", 288, 384, True, 6), ("psn", 288, 384, True, 4), ("lif", 288, 384, True, 10), ("lif", 256, 320, False, 3), ("psn", 160, 224, False, 5), ("lif", 160, 256, False, 4)])
def test_replica_forward_is_bit_equal_to_separate_batch1_forwards(kind, H, W, en4, R):
    """BASELINE configs[1] (en4, 288 x 384) and the 3-encoder model: R samples through forward_replicas = R forwards of one sample,
    every flow map bit for bit (the products are exact integer sums of digits / fp32 epilogues per element: no result depends on
    which other rows a launch carries) - and NOT the reference's batch-of-R forward, which couples the samples."""
    from sdformerflow_amd.harness import prepare_chunk
    from sdformerflow_amd.synthetic import synth_voxel
    model = _model(kind, H, W, en4)
    xs = [prepare_chunk(synth_voxel(1, 10, H, W, seed=300 + 7 * i)).to(DEV) for i in range(R)]
    with torch.no_grad():
        ones = [[f.clone() for f in model(x)["flow"]] for x in xs]
        rep = model.forward_replicas(torch.cat(xs, 0))["flow"]
        bat = model(torch.cat(xs, 0))["flow"]
    torch.cuda.synchronize()
    assert len(rep) == len(ones[0])
    for lvl, f in enumerate(rep):
        assert f.shape == (R, 2, H, W)
        for i in range(R):
            assert torch.equal(f[i], ones[i][lvl][0]), (lvl, i, (f[i] - ones[i][lvl][0]).abs().max().item())
    if (H, W) in ((288, 384), (144, 192), (160, 256)):                  # (the other geometries have an odd window count per sample at some stage:
        assert not torch.equal(bat[-1], rep[-1])                        #  forward_replicas serves them one by one - the reference's batch view differs either way)
    # and a second call (cached tables, no tape) reproduces it; then a plain forward still runs on its own tables
    with torch.no_grad():
        again = model.forward_replicas(torch.cat(xs, 0))["flow"]
        one = model(xs[0])["flow"]
    assert all(torch.equal(a, b) for a, b in zip(again, rep))
    assert all(torch.equal(a, b) for a, b in zip(one, ones[0]))


@pytest.mark.gpu
def test_replica_forward_refuses_the_tape():
    model = _model("lif", 144, 192, False)
    eng = model.engine()
    eng.tape = []
    try:
        with pytest.raises(hip.SdfError):
            eng.forward(torch.zeros((2, 10, 2, 144, 192), device=DEV), None, replicas=True)
    finally:
        eng.tape = None


@pytest.mark.gpu
@pytest.mark.parametrize("ns", [2, 3])
@pytest.mark.parametrize("nW,nH,R,rows_big", [(4, 3, 3, False), (2, 6, 2, False), (10, 3, 40, True)])
def test_head_scramble_of_replicas_at_the_kernel(nW, nH, R, rows_big, ns):
    """sdf_spike_gemm_fwd with zg_rep: the projection's operand gather of R independent problems in one (T', R nW, N1, C) buffer = R
    separate calls on every replica's own (T', nW, N1, C) tensor, bit for bit (same tile kernel, same K order) - both the small-tile
    kernel and, at many rows, whatever the dispatcher picks."""
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    Tq, N1 = 2, 81
    Cc = nH * 32
    g = torch.Generator().manual_seed(7 + nW)
    E = (torch.rand((Tq, R, nW, N1, Cc), generator=g) < 0.3).to(torch.uint8)
    W = hip.split_weight(rnd((Cc, Cc), 41, -0.3, 0.3).to(DEV), ns)
    M1 = Tq * nW * N1
    out = torch.empty((Tq * R * nW * N1, Cc), device=DEV)
    hip.spike_gemm(E.to(DEV), W, out, Tq * R * nW * N1, Cc, Cc, zg=(nH, Tq, R * nW, N1, nW))
    out = out.view(Tq, R, nW * N1, Cc)
    for r in (range(R) if not rows_big else (0, R // 2, R - 1)):
        Er = E[:, r].contiguous().to(DEV)
        o1 = torch.empty((M1, Cc), device=DEV)
        hip.spike_gemm(Er, W, o1, M1, Cc, Cc, zg=(nH, Tq, nW, N1))
        assert torch.equal(out[:, r].reshape(M1, Cc), o1), r
    with pytest.raises(hip.SdfError):                                      # windows per replica must divide B_ and be a multiple of T'
        hip.spike_gemm(E.to(DEV), W, out.view(-1, Cc), Tq * R * nW * N1, Cc, Cc, zg=(nH, Tq, R * nW, N1, 3 if nW != 3 else 5))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_neuron_descriptor_outermost_dimension(kind):
    """SdfNeuronDesc.nrep: the batch elements of a pixel-strided channel slice as ONE descriptor = one launch per batch element."""
    from sdformerflow_amd.synthetic import synth_uniform as rnd
    B, T, hw, C2, cp, c1 = 3, 10, 35, 96, 208, 16
    x = rnd((B, T, hw, C2), 5, -0.4, 0.8).to(DEV)
    p = hip.NeuronParams("psn", psn_w=rnd((T, T), 6, -0.3, 0.6).to(DEV), psn_b=rnd((T,), 7, -0.2, 0.1).to(DEV)) if kind == "psn" else \
        hip.NeuronParams("lif", 2.0, 0.1, None)
    one = torch.zeros((B, T, hw, cp), dtype=torch.uint8, device=DEV)
    hip.neuron_multi_fwd([(x, one.view(-1)[c1:], T, hw, C2, C2, hw * C2, cp, hw * cp, p, None, 0, None, None, 0, 1, None, 0, 0, None,
                           (B, T * hw * C2, T * hw * cp)),
                          (x, one.view(-1)[c1 + C2:], T, hw, 32, C2, hw * C2, cp, hw * cp, p, None, 0, None, None, 0, 1, None, 0, 0, None,
                           (B, T * hw * C2, T * hw * cp))])
    ref = torch.zeros_like(one)
    for b in range(B):
        hip.neuron_fwd(x[b], ref[b].view(-1)[c1:], T, hw, C2, C2, hw * C2, cp, hw * cp, p)
        hip.neuron_fwd(x[b], ref[b].view(-1)[c1 + C2:], T, hw, 32, C2, hw * C2, cp, hw * cp, p)
    assert torch.equal(one, ref) and 0.02 < one[..., c1:c1 + C2].float().mean() < 0.98


@pytest.mark.gpu
def test_launch_log_reports_every_launch_of_a_call():
    """sdf_launch_log: kernel name, workgroups, threads and an event-timed duration per launch; off again afterwards."""
    x = torch.rand((10, 1 << 16), device=DEV)
    with hip.launch_log() as log:
        hip.lif_fwd(x, 2.0, 0.1, None, torch.uint8)
        hip.lif_fwd(x, 2.0, 0.1, None, torch.uint8)
    assert len(log.rows) == 2
    for name, wgs, thr, lds, us in log.rows:
        assert "neuron_kernel<10>" in name and wgs == (1 << 16) // 4 // 256 and thr == 256 and 0.0 < us < 1e4
    with hip.launch_log() as log2:
        pass
    assert log2.rows == []
    hip.lif_fwd(x, 2.0, 0.1, None, torch.uint8)                        # (not logged: the log is off)
    with hip.launch_log() as log3:
        pass
    assert log3.rows == []


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["lif", "psn"])
def test_replica_forward_at_configs4_shape_T20(kind):
    """BASELINE configs[4]'s shape (20 bins / T = 20, 480 x 640): its kernels are other instantiations (the digit convolution's rolled
    time loop, T = 20 row-loop and K-ring tiles, sample chunks where a launch would pass its 31-bit offsets) - two replicas are two
    batch-1 forwards there too, bit for bit."""
    from sdformerflow_amd.harness import prepare_chunk
    from sdformerflow_amd.synthetic import synth_voxel
    model = _model(kind, 480, 640, True, T=20)
    xs = [prepare_chunk(synth_voxel(1, 20, 480, 640, seed=500 + i)).to(DEV) for i in range(2)]
    with torch.no_grad():
        ones = [[f.clone() for f in model(x)["flow"]] for x in xs]
        rep = model.forward_replicas(torch.cat(xs, 0))["flow"]
    torch.cuda.synchronize()
    for lvl, f in enumerate(rep):
        for i in range(2):
            assert torch.equal(f[i], ones[i][lvl][0]), (lvl, i, (f[i] - ones[i][lvl][0]).abs().max().item())


@pytest.mark.gpu
def test_forward_replicas_serves_what_it_cannot_batch_one_by_one():
    """The same call on models the replica tables / digit kernels do not cover - the MDR config (T = 5: layers on the streaming kernels),
    the SEW family (its engine keeps the reference's batch view) and the exact 3-plane weight mode - returns the flows of separate
    batch-1 forwards as well: it runs them one after the other."""
    import os
    import yaml
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet_en4, SpikingformerFlowNet
    from sdformerflow_amd.harness import prepare_chunk
    from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cdir = os.path.join(root, "sdformerflow_amd", "configs")

    def load(m):
        m.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}), strict=True)
        return m.eval().to(DEV)
    cfg = yaml.safe_load(open(os.path.join(cdir, "train_MDR_supervised_SDformerFlow.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"])
    cfg["swin_transformer"]["input_size"] = [256, 256]
    mdr = load(MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy()))
    cfg = yaml.safe_load(open(os.path.join(cdir, "train_DSEC_supervised_SDformerFlow_en4.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type="lif")
    cfg["swin_transformer"].update(input_size=[144, 192], swin_depths=[2, 2, 6], swin_num_heads=[3, 6, 12], swin_out_indices=[0, 1, 2])
    sew = load(SpikingformerFlowNet(cfg["model"].copy(), cfg["swin_transformer"].copy()))
    exact = _model("lif", 144, 192, False)
    exact.gemm_nsplit = 3
    for model, T, H, W in ((mdr, mdr.engine().num_bins, 256, 256), (sew, 10, 144, 192), (exact, 10, 144, 192)):     # (T: voxel bins)
        xs = [prepare_chunk(synth_voxel(1, T, H, W, seed=700 + i)).to(DEV) for i in range(2)]
        with torch.no_grad():
            ones = [[f.clone() for f in model(x)["flow"]] for x in xs]
            rep = model.forward_replicas(torch.cat(xs, 0))["flow"]
        for lvl, f in enumerate(rep):
            for i in range(2):
                assert torch.equal(f[i], ones[i][lvl][0]), (type(model).__name__, lvl, i)
