#!/usr/bin/env python3
"""Benchmark of the SDformerFlow forward hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--neuron lif|psn] [--no-cpu] [--inflight F] [--replicas R] [--eager]
    python bench.py --train [--local-batch B] ...      (BASELINE configs[3], the training step; not the default workload)

One "step" = one forward of MS_SpikingformerFlowNet_en4 over one synthetic 1 x 10 x 2 x 288 x 384 event
voxel (BASELINE config 2) already resident in HBM.  For N > 1 the driver launches one process per GPU
(torch.distributed.run); the forward does not shard inside a micro-batch (batch elements are
coupled by the reference's raw reshapes), so every rank runs an independent replica: weak scaling, no
data-path collective, value = N*K / max-over-ranks(time).

Every rank keeps F (default 2) HIP streams busy, each replaying the HIP graph of ONE launch sequence that carries R (default 10)
independent batch-1 forwards of different voxels (`model.forward_replicas`: every sample keeps the reference's batch-1
semantics - a batch of R would couple the samples - and its flow maps are asserted bit-equal to a separate forward in this
run; `--eager` launches kernel by kernel instead): the layers of a batch-1 forward are too small to fill 256 CUs one
kernel at a time and spend a fixed few microseconds per launch on cold operands and weight staging, R samples share both.
A step is still one whole forward of one sample; the K timed steps are dealt evenly to the streams (class InFlight).  The
single-stream latency of a plain forward is reported next to the throughput.

Prints ONE JSON line on rank 0 with `roofline` (dominant own kernel, timed live with HIP events on the
launch stream) and `cpu_baseline` (the CPU oracle = port of the reference, timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_DENSE_TFLOPS = 2500.0     # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_HBM_GBPS = 8000.0              # same guide: 8.0 TB/s spec (6.3 TB/s achievable)
# HBM bytes per launch of the roofline shape from the PMC counters.  They cannot be read inside this process (rocprofv3 owns the
# counters), so the figure is the one measured by tools/pmc_traffic.sh on the build named in CONV_TRAFFIC_SOURCE
# (round 5: profiles/r5u_pmc_conv_mapping.txt, re-measured on this round's tree; both epilogue forms, since `roofline` averages over them)
CONV_TRAFFIC_KIB = {"membrane_and_spikes": (68172.0, 133.6e6 / 1024), "spikes_only": (15397.2, 26.5e6 / 1024)}    # (FETCH_SIZE, WRITE_SIZE) KiB / launch
CONV_TRAFFIC_BYTES = sum(2 * f + w for f, w in CONV_TRAFFIC_KIB.values()) / 2 * 1024                 # FETCH_SIZE x 2: gfx950 correction
CONV_ALGORITHMIC_BYTES = {"membrane_and_spikes": 10 * 144 * 192 * 96 * (1 + 4 + 4 + 1) + 3 * 96 * 864,
                          "spikes_only": 10 * 144 * 192 * 96 * (1 + 1) + 3 * 96 * 864}
CONV_TRAFFIC_SOURCE = ("NOT measured in this run (rocprofv3 owns the counters): profiles/r5u_pmc_conv_mapping.txt (tools/pmc_conv_ab.sh on the "
                       "round-5 tree: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes on this exact launch, gfx950 x2 read "
                       "correction, KiB = 1024 B): membrane + spikes form 139.6 MB read + 133.6 MB written against 265.7 MB algorithmic "
                       "(1.03x), spikes-only form 31.5 + 26.5 MB against 53.4 MB (1.09x); the four launches inside one forward: 341.6 MB read + "
                       "320.9 MB written (profiles/r5q_pmc_forward.txt) = 165.6 MB per launch.  The three workgroups that serve the column blocks "
                       "of a tile range now sit on one XCD and walk it side by side (the spike image leaves HBM once; round 2's column-block-"
                       "major ranges read it 3x: 196.9 / 88.7 MB, same file); `traffic` is the mean of the two forms, as `achieved` is the "
                       "mean over the forward's two launches of each form")


def build_model(kind, device):
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet_en4
    from sdformerflow_amd.synthetic import synth_state_dict
    cfg = yaml.safe_load(open(os.path.join(ROOT, "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type=kind)
    cfg["swin_transformer"]["input_size"] = [288, 384]
    model = MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy())
    sd = synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict(sd, strict=True)
    model.eval()
    return model.to(device), sd


def synthetic_chunk(seed=1235):
    """Harness input prep of eval_DSEC_flow_SNN.py:179-205 on a seeded DSEC-like voxel."""
    from sdformerflow_amd.harness import prepare_chunk
    from sdformerflow_amd.synthetic import synth_voxel
    return prepare_chunk(synth_voxel(1, 10, 288, 384, seed=seed))


def _timed(fn, sets, iters):
    """Average launch time of fn(set) cycling through `sets`, HIP events on the launch stream (= torch's current stream,
    the one passed through the C ABI)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(max(3, len(sets))):
        fn(sets[i % len(sets)])
    e0.record()
    for i in range(iters):
        fn(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


L3_BYTES = 256 << 20                # Infinity Cache of the MI355X (MI355X_MICROARCH.md): replaying ONE working set smaller than
                                    # this measures the L3-resident rate, not HBM - both figures are reported


def time_dominant_kernels(model, iters=40, R=1):
    """Live HIP-event timing of the dominant kernel of the forward - the 3x3 spike convolution of the patch embedding's
    res-blocks (10 images of 144x192, 96 -> 96 channels) in the form the forward launches it: BN + identity stored as the fp32
    membrane AND the next block's LIF over T=10 on it, one launch (MS_ResBlock conv2 + sn1; 2 of the forward's 5 big 3x3
    launches, the other 3 differ only in the epilogue) - and of the HBM-bound neuron update.  Each is timed twice: rotating
    through enough operand sets that consecutive launches never find their operands in the 256 MiB Infinity Cache (> 2 x L3
    between two uses of a set: the HBM figure, the one in `achieved` / `frac`), and replaying one set (`l3_resident`, what
    round 1 reported).  The plain fp32-epilogue launch of the same convolution (round 1's roofline shape) is reported beside it."""
    from sdformerflow_amd import hip
    eng = model.engine()
    dev = eng.device
    imgs, H, W, Cc = 10, 144, 192, 96
    rb = eng.pe_res[0]
    set_bytes = imgs * H * W * Cc * (1 + 4 + 4 + 1)
    nset = -(-3 * L3_BYTES // set_bytes)                          # 265 MB per set -> 4 sets in rotation
    sets = [((torch.rand((1, imgs, H, W, Cc), device=dev) < 0.3).to(torch.uint8), torch.rand((1, imgs, H, W, Cc), device=dev))
            for _ in range(nset)]
    sn = eng.pe_res[1].sn1

    def conv_fusedm(st):                                          # res-block conv2: membrane + the next block's spikes
        eng._conv3x3(st[0], rb.w2, Cc, bn=rb.bn2, resid=st[1], sn=sn, membrane=True)

    def conv_fused(st):                                           # res-block conv1: BN -> LIF spikes only
        eng._conv3x3(st[0], rb.w1, Cc, bn=rb.bn1, sn=rb.sn2)

    def conv_f32(st):
        eng._conv3x3(st[0], rb.w2, Cc, bn=rb.bn2, resid=st[1])
    t_m, t_m_l3 = _timed(conv_fusedm, sets, iters), _timed(conv_fusedm, sets[:1], iters)
    t_s, t_s_l3 = _timed(conv_fused, sets, iters), _timed(conv_fused, sets[:1], iters)
    # the forward launches this kernel four times on this shape: twice in each form -> its average launch duration
    t_conv, t_conv_l3 = (t_m + t_s) / 2, (t_m_l3 + t_s_l3) / 2
    t_f32 = _timed(conv_f32, sets, iters)
    t_f32_l3 = _timed(conv_f32, sets[:1], iters)
    del sets
    # the same two launches as the headline's scheme makes them: R samples per launch sequence = 10 R images per launch (one set is
    # R x 265 MB: beyond the Infinity Cache by itself from R = 2 on; two sets in rotation)
    tR = None
    if R > 1:
        torch.cuda.empty_cache()
        setsR = [((torch.rand((R, imgs, H, W, Cc), device=dev) < 0.3).to(torch.uint8), torch.rand((R, imgs, H, W, Cc), device=dev)) for _ in range(2)]
        eng.replicas = True                                       # (routing as in forward_replicas)
        try:
            tR = (_timed(conv_fusedm, setsR, max(6, iters // 4)), _timed(conv_fused, setsR, max(6, iters // 4)))
        finally:
            eng.replicas = False
        del setsR
        torch.cuda.empty_cache()
    flops = 2.0 * imgs * H * W * Cc * 9 * Cc                    # algorithmic: one multiply-add per (pixel, cout, tap, cin)
    digits = getattr(rb.w2, "digits", None) is not None and sn.kind != "psn"
    ns = int(rb.w2.shape[0])
    issued = 1.5 if digits else float(ns)                        # MFMA work per algorithmic flop on the 16-bit pipe's scale
    kname = ("sdfmm::spike_conv_wres_i8_kernel<10,6,1,3> (weights resident in LDS as 3 int8 digit planes, halo tiles, LIF over T fused, "
             "3 wave groups per workgroup)"
             if digits else f"sdfmm::spike_mm_pp_kernel<{ns},10,true> (streaming ping-pong kernel)")
    gemm = {"kernel": kname + " - 3x3 spike conv 96->96 @ 10x144x192; average over the forward's four launches of it on this shape "
                              "(2 x BN -> LIF(T=10) spikes, 2 x BN + identity -> fp32 membrane + LIF spikes)",
            "bound": "mfma", "achieved": flops / t_conv / 1e12, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
            "us_per_launch": t_conv * 1e6, "operand_sets_in_rotation": nset, "rotation_bytes": nset * set_bytes,
            "l3_resident": {"us_per_launch": t_conv_l3 * 1e6, "achieved": flops / t_conv_l3 / 1e12,
                            "frac": flops / t_conv_l3 / 1e12 / PEAK_BF16_DENSE_TFLOPS},
            "forms": {"spikes_only": {"us_per_launch": t_s * 1e6, "frac": flops / t_s / 1e12 / PEAK_BF16_DENSE_TFLOPS,
                                      "l3_resident_us_per_launch": t_s_l3 * 1e6},
                      "membrane_and_spikes": {"us_per_launch": t_m * 1e6, "frac": flops / t_m / 1e12 / PEAK_BF16_DENSE_TFLOPS,
                                              "l3_resident_us_per_launch": t_m_l3 * 1e6}},
            "fp32_epilogue_form": {"us_per_launch": t_f32 * 1e6, "achieved": flops / t_f32 / 1e12,
                                   "frac": flops / t_f32 / 1e12 / PEAK_BF16_DENSE_TFLOPS, "l3_resident_us_per_launch": t_f32_l3 * 1e6,
                                   "note": "the same convolution with the plain BN + residual fp32 epilogue (round 1's roofline shape)"},
            "algorithmic_bytes": sum(CONV_ALGORITHMIC_BYTES.values()) / 2,
            "traffic": CONV_TRAFFIC_BYTES if digits else None, "traffic_unit": "bytes per launch (HBM read + write), mean of the two forms",
            "traffic_over_algorithmic": CONV_TRAFFIC_BYTES / (sum(CONV_ALGORITHMIC_BYTES.values()) / 2) if digits else None,
            "traffic_source": CONV_TRAFFIC_SOURCE,
            "pmc": "profiles/r5q_pmc_forward.txt (the four launches inside one forward of the round-5 tree: 5.7 VALU per MFMA, pipe busy 40.0 %, LDS conflicts 5.9 %), r4e_pmc_forward.txt, r3s_pmc_forward.txt and r3i_pmc_kernels_after_operand_swap.txt: 683 MFMA "
                   "and ~4 000 VALU instructions per wave (6 VALU per MFMA; 8.2 before the weights became the MFMA's row operand, "
                   "r3g_pmc_kernels.txt), SQ_VALU_MFMA_BUSY_CYCLES = 39 % of the kernel's cycles per SIMD at an effective 2.2 GHz, LDS bank "
                   "conflicts 6 % of LDS cycles: the kernel is bound by its epilogue's vector instructions, not by the matrix pipe",
            "note": "algorithmic flops (2 per multiply-add of the convolution) against the dense bf16/f16 MFMA peak.  The kernel issues "
                    + ("3 int8 digit MFMAs (v_mfma_i32_32x32x32_i8, K = 32 in the cycles the 16-bit form needs for K = 16) per product: 1.5x "
                       "the algorithmic work on the 16-bit pipe's scale" if digits else f"{ns} 16-bit MFMAs per product") +
                    "; the matrix pipe is busy 40 % of the kernel's cycles (PMC, profiles/r5q_pmc_forward.txt)"}
    gemm["frac"] = gemm["achieved"] / gemm["peak"]
    gemm["frac_of_issued_mfma"] = issued * gemm["frac"]
    if tR is not None and digits:
        # `achieved` / `frac` / `us_per_launch` describe the launch the TIMED REGION makes (R samples = 10 R images: the dispatcher takes two
        # row blocks per wave at this row count, spike_conv_wres_i8_kernel<10,6,2,2>); the one-sample launch of rounds 2 - 5 stays beside it
        one = {k: gemm[k] for k in ("achieved", "frac", "frac_of_issued_mfma", "us_per_launch", "forms", "l3_resident")}
        tc = (tR[0] + tR[1]) / 2
        gemm.update(achieved=R * flops / tc / 1e12, us_per_launch=tc * 1e6, images_per_launch=imgs * R, samples_per_launch=R,
                    one_sample_per_launch=one,
                    forms={"spikes_only": {"us_per_launch": tR[1] * 1e6, "frac": R * flops / tR[1] / 1e12 / PEAK_BF16_DENSE_TFLOPS},
                           "membrane_and_spikes": {"us_per_launch": tR[0] * 1e6, "frac": R * flops / tR[0] / 1e12 / PEAK_BF16_DENSE_TFLOPS}})
        gemm.pop("l3_resident", None)
        gemm["frac"] = gemm["achieved"] / gemm["peak"]
        gemm["frac_of_issued_mfma"] = issued * gemm["frac"]
        gemm["algorithmic_bytes"] = R * gemm["algorithmic_bytes"]
        gemm["kernel"] = (f"sdfmm::spike_conv_wres_i8_kernel (3 int8 digit planes resident in LDS, halo tiles, LIF over T fused) - 3x3 spike conv 96->96 on {imgs * R} "
                          f"images of 144x192 = the launch of the headline's scheme ({R} samples per launch sequence; <10,6,2,2>: two row blocks per wave "
                          "from 7 samples on, <10,6,1,3> below); average over the forward's four launches of it (2 x BN -> LIF(T=10) spikes, 2 x BN + "
                          "identity -> fp32 membrane + LIF spikes); `one_sample_per_launch`: the 10-image launch of rounds 2 - 5")
    # round 6: the traffic of this kernel's four launches inside the headline's launch sequence, from the counters tools/pmc_forward2.sh
    # took on this round's tree (profiles/r6_kernel_counters.json; rocprofv3 owns the counters, they cannot be read in this process)
    try:
        cj = json.load(open(os.path.join(ROOT, "profiles", "r6_kernel_counters.json")))
        ks = [v for k, v in cj["per_sample"].items() if k.startswith("spike_conv_wres_i8_kernel<10, 6, ")]      # (the stride-1 96-channel forms: <10,6,1,3,..> / <10,6,2,2,..>)
        if digits and ks:
            per_launch = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in ks) / sum(v["launches"] for v in ks) * gemm.get("samples_per_launch", 1)
            gemm.update(traffic=per_launch, traffic_over_algorithmic=per_launch / gemm["algorithmic_bytes"],
                        traffic_source="profiles/r6_kernel_counters.json (tools/pmc_forward2.sh on the round-6 tree: rocprofv3 --pmc FETCH_SIZE and --pmc "
                                       "WRITE_SIZE in separate passes over one eager launch sequence of " + str(cj["samples_per_launch_sequence"]) +
                                       " samples, gfx950 x 2 read correction, KiB = 1024 B; bytes of this kernel's launches / (launches x samples): "
                                       "the mean over its two epilogue forms, as `achieved` is); the round-5 figure from the isolated launches: " +
                                       f"{CONV_TRAFFIC_BYTES:.0f} (profiles/r5u_pmc_conv_mapping.txt)")
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        pass
    # neuron: T=10 over the stage-0 MLP hidden tensor shape (10 x 72*96*384 fp32 in, u8 out)
    blk = eng.stages[0][0]
    n = 72 * 96 * 384
    nset = -(-3 * L3_BYTES // (10 * n * 5))                       # 133 MB per set -> 7 sets
    sets = [(torch.rand((10, n), device=dev) - 0.3, torch.empty((10, n), dtype=torch.uint8, device=dev)) for _ in range(nset)]
    p = blk.sn2

    def neur(st):
        hip.neuron_fwd(st[0], st[1], 10, 1, n, 0, n, 0, n, p)
    t_n = _timed(neur, sets, iters)
    t_n_l3 = _timed(neur, sets[:1], iters)
    del sets
    neuron = {"kernel": "neuron_kernel<10> (26.5 M neurons x T=10, f32 in / u8 out)", "bound": "hbm",
              "achieved": 10.0 * n * 5 / t_n / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "us_per_launch": t_n * 1e6,
              "operand_sets_in_rotation": nset, "rotation_bytes": nset * 10 * n * 5,
              "l3_resident": {"us_per_launch": t_n_l3 * 1e6, "achieved": 10.0 * n * 5 / t_n_l3 / 1e9,
                              "frac": 10.0 * n * 5 / t_n_l3 / 1e9 / PEAK_HBM_GBPS},
              "traffic": None}
    neuron["frac"] = neuron["achieved"] / neuron["peak"]
    return gemm, neuron


def streams_rate(fwd, inputs, iters):
    """Seconds per call of fwd(x) with len(inputs) calls in flight: every input gets its own HIP stream and the HIP graph of one call,
    the graphs are replayed round-robin (the headline's streams-and-graphs level for the side configurations, whose batches keep the
    reference's own batch semantics).  Raises when a forward cannot be captured."""
    streams = [torch.cuda.Stream() for _ in inputs]
    graphs, outs = [], []
    for st, x in zip(streams, inputs):
        with torch.cuda.stream(st):
            for _ in range(2):
                fwd(x)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            outs.append(fwd(x))
        graphs.append(g)
    torch.cuda.synchronize()
    for i in range(len(inputs)):
        with torch.cuda.stream(streams[i]):
            graphs[i].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        with torch.cuda.stream(streams[i % len(inputs)]):
            graphs[i % len(inputs)].replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    del graphs
    return dt, outs


def time_config3(iters=10):
    """BASELINE configs[2] beside the headline: the ANN STTFlowNet (STT_voxel config: 20 bins, patch (10,4,4), window (2,9,9)) at
    batch 8, 288 x 384, synthetic weights and voxels - ms per batch, samples/s - and its dominant kernel, the dense 3x3
    convolution of the patch embedding on two fp16 planes per operand (csrc/dense_conv_wres.hip), timed with HIP events on the
    launch stream on config 3's own shape (16 images x 96 channels x 288 x 384, BatchNorm + residual + ReLU in the epilogue)."""
    import yaml
    from sdformerflow_amd import hip
    from sdformerflow_amd.STSwinNet import STSwinNet
    from sdformerflow_amd.synthetic import synth_state_dict, synth_voxel
    B, H, W = 8, 288, 384
    cfg = yaml.safe_load(open(os.path.join(ROOT, "sdformerflow_amd", "configs", "train_DSEC_supervised_STT_voxel.yml")))
    net = STSwinNet.STTFlowNet(dict(cfg["model"], spiking_neuron=None), dict(cfg["swin_transformer"], input_size=[H, W])).eval()
    skip = ("relative_position_index", "relative_coords_table", "num_batches_tracked")
    net.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items() if not k.endswith(skip)}), strict=False)
    net = net.to("cuda")
    vox = synth_voxel(B, 20, H, W, seed=1237).to("cuda")
    with torch.no_grad():
        for _ in range(3):
            flow = net(vox, None)["flow"]
        assert all(torch.isfinite(f).all() for f in flow)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            net(vox, None)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        # the same forward with two batches in flight (own stream, own HIP graph each): what the headline's scheme gives this configuration
        two = None
        try:
            vox2 = synth_voxel(B, 20, H, W, seed=1247).to("cuda")
            dt2, outs2 = streams_rate(lambda v: net(v, None)["flow"], [vox, vox2], 2 * iters)
            # (config 3 is not bit-equal run to run: the library convolutions of its 384-channel res-blocks pick their reduction order)
            assert all(float((a - b).abs().max()) <= 1e-4 * float(b.abs().max()) for a, b in zip(outs2[0], flow)), "a graph replay of config 3 left the eager forward"
            two = {"samples_per_s": B / dt2, "ms_per_batch": dt2 * 1e3, "in_flight": 2, "hip_graph": True}
            del outs2, vox2
        except Exception as e:                                       # (a side figure: never takes the line down)
            two = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    g = torch.Generator().manual_seed(0)
    sets = []
    for _ in range(2):                                             # 2 x (0.68 GB in + 0.68 GB residual + 0.68 GB out) > 3 x the Infinity Cache
        xp = hip.pack_planes(torch.randn(2 * B, 96, H, W, generator=g).cuda())
        rp = hip.pack_planes(torch.randn(2 * B, 96, H, W, generator=g).cuda())
        sets.append((xp, rp))
    wp = hip.pack_dense_conv_weight((torch.randn(96, 96, 3, 3, generator=g) / 30).cuda())
    al, be = (0.5 + torch.rand(96, generator=g)).cuda(), torch.randn(96, generator=g).cuda()
    t = _timed(lambda s_: hip.dense_conv3x3(s_[0], wp, al, be, s_[1], True), sets, 20)
    flop = 2 * B * H * W * 96 * 864 * 2
    del sets
    # the attention half of a first-stage block (704 windows of 162 tokens, C = 96): the one-launch kernel against the four launches it
    # replaces (LayerNorm, qkv Linear, window attention, proj Linear), HIP events on the launch stream, four input sets in rotation
    half = {}
    try:
        from sdformerflow_amd.STSwinNet.swin_transformer3D_v2 import SwinTransformerBlock3D, compute_mask, layer_norm
        blocks = [m for m in net.modules() if isinstance(m, SwinTransformerBlock3D) and m.dim == 96]
        for name, blk in (("plain", blocks[0]), ("shifted", blocks[1])):
            ws, ss = blk.window_size, blk.shift_size
            D_, H_, W_ = 2, 72, 96
            row_map, B_ = hip.window_slice_map(B, D_, H_, W_, ws, ss, "cuda")
            pad = [d + (-d) % w for d, w in zip((D_, H_, W_), ws)]
            mask = compute_mask(*pad, ws, ss, torch.device("cuda")).contiguous() if any(ss) else None
            xs = [torch.randn(B * D_ * H_ * W_, 96, generator=g).cuda() for _ in range(4)]
            with torch.no_grad():
                t1 = _timed(lambda x2: blk.attn.forward_rows(None, row_map, B_, mask, x2, norm=blk.norm1), xs, 20)
                t4 = _timed(lambda x2: blk.attn.forward_rows(layer_norm(blk.norm1, x2), row_map, B_, mask, x2), xs, 20)
            fl = B_ * (2 * 162 * 96 * 288 + 4 * 162 * 162 * 32 * 3 + 2 * 162 * 96 * 96)
            half[name] = {"one_launch_us": t1 * 1e6, "four_launches_us": t4 * 1e6, "windows": B_, "algorithmic_gflop": fl / 1e9,
                          "tflops": fl / t1 / 1e12, "frac_of_2500": fl / t1 / 1e12 / 2500.0}
        # the MLP half of the same block: LayerNorm -> fc1 -> GELU -> fc2 -> + x, one launch (csrc/ann_mlp_block.hip) against three
        blk = blocks[0]
        xs = [torch.randn(B * 2 * 72 * 96, 96, generator=g).cuda() for _ in range(4)]
        with torch.no_grad():
            pk = hip.pack_ann_mlp_block_weights(blk.mlp.fc1.weight, blk.mlp.fc2.weight)
            n2 = blk.norm2
            t1 = _timed(lambda x2: hip.ann_mlp_block(x2, torch.empty_like(x2), n2.weight.detach(), n2.bias.detach(), n2.eps, pk[0], blk.mlp.fc1.bias.detach(),
                                                     pk[1], blk.mlp.fc2.bias.detach()), xs, 20)
            t3 = _timed(lambda x2: blk.mlp(layer_norm(n2, x2), x2), xs, 20)
        half["mlp_half"] = {"one_launch_us": t1 * 1e6, "three_launches_us": t3 * 1e6, "rows": xs[0].shape[0],
                            "note": "x + fc2(GELU(fc1(LayerNorm(x)))), C = 96, hidden 384: the hidden activations (170 MB) never reach HBM"}
        half["note"] = ("x + proj(window attention(LayerNorm(x))) of BASELINE configs[2]'s first stage (8 x 2 x 72 x 96 tokens, C = 96, three heads): "
                        "csrc/ann_block.hip (q | k | v never in HBM) against sdf_layer_norm_fwd + sdf_dense_linear_fwd + sdf_win_attn_fwd + "
                        "sdf_dense_linear_fwd; matrix pipe busy 24.7 % (profiles/r5z_pmc_ann_block.txt)")
    except Exception as e:                                         # a side line must never take the headline down with it
        half = {"error": repr(e)[:300]}
    return {"workload": "BASELINE configs[2]: STTFlowNet (ANN) forward, batch 8, 20-bin 288x384 voxel, fp32 activations", "samples_per_s": B / dt,
            "two_batches_in_flight": two, "attention_half_block_stage0": half,
            "ms_per_batch": dt * 1e3, "dtype": "f32 as f16x2 (hi + lo planes of both operands, three products, fp32 accumulate)",
            "roofline": {"kernel": "dense_conv_wres_kernel<6> (16 x 96 x 288 x 384, 3x3, BN + residual + ReLU fused)", "bound": "mfma",
                         "achieved": flop / t / 1e12, "executed_on_pipe": 3 * flop / t / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": flop / t / 1e12 / 2500.0, "frac_executed": 3 * flop / t / 1e12 / 2500.0, "us_per_launch": t * 1e6,
                         "algorithmic_flop_per_launch": flop, "operand_sets_in_rotation": 2, "traffic": 1.714e9 + 0.680e9,
                         "traffic_over_algorithmic": (1.714e9 + 0.680e9) / 2.04e9,
                         "traffic_note": "HBM bytes per launch of the shipped dense_conv_wres_kernel<6, 2>, NOT measured in this run (rocprofv3 owns the "
                                         "counters): profiles/r4e_pmc_dense_conv.txt (tools/pmc_dense_conv.sh, round 4: FETCH_SIZE 837 023 KiB x 2 on gfx950 "
                                         "+ WRITE_SIZE 663 560 KiB, separate --pmc passes, program directly behind `--`); algorithmic 2.04e9 (round 2's "
                                         "kernel: 2.61e9, profiles/r2p_pmc_dense.txt)"}}


class InFlight:
    """The headline's execution scheme: F streams, each replaying a HIP graph of ONE launch sequence that carries R independent
    batch-1 forwards ("replicas", model.forward_replicas: every sample keeps the reference's batch-1 semantics and its flows are
    bit-equal to a plain forward of that sample - checked by `verify`; R = 1 is the plain forward).  A step is one forward of one
    sample: run(n) issues exactly n of them - dealt evenly to the streams, each stream's share as replays of its R-sample graph plus
    one replay of a graph captured for the remainder (`_plan`)."""

    def __init__(self, model, dev, F_, R, eager=False, seed0=1235):
        self.model, self.dev, self.F, self.R, self.eager = model, dev, F_, R, eager
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(F_)]
        self.seed0 = seed0
        self._voxels = {}
        self.slots = {}                                          # (stream index, batch) -> [input, output, graph]
        for j in range(F_):
            self._slot(j, R)
        torch.cuda.synchronize()

    def voxel(self, k):
        """Sample k of the synthetic stream (seed seed0 + k), on the device; every (stream, position in the batch) has its own."""
        if k not in self._voxels:
            self._voxels[k] = synthetic_chunk(self.seed0 + k).to(self.dev)
        return self._voxels[k]

    def _fwd(self, x):
        return self.model.forward_replicas(x) if x.shape[0] > 1 else self.model(x)

    def _slot(self, j, n):
        if (j, n) not in self.slots:
            x = torch.cat([self.voxel(j * self.R + b) for b in range(n)], 0)
            st = self.streams[j]
            with torch.cuda.stream(st):
                for _ in range(2):
                    o = self._fwd(x)                             # also creates this stream's split-K workspace and plan caches
            torch.cuda.synchronize()
            g = None
            if not self.eager:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    o = self._fwd(x)
            self.slots[(j, n)] = [x, o, g]
        return self.slots[(j, n)]

    def _plan(self, n):
        """n samples dealt as evenly as possible to the F streams; a stream's share runs as replays of its R-sample graph plus one
        replay of a graph captured for the remainder: [(stream, samples)] in issue order (round-robin over the streams)."""
        if getattr(self, "plan_override", None) and sum(m for _, m in self.plan_override) == n:
            return list(self.plan_override)                      # (diagnostic: --plan "0:10,1:5,1:5")
        share = [n // self.F + (1 if j < n % self.F else 0) for j in range(self.F)]
        per = [[self.R] * (s // self.R) + ([s % self.R] if s % self.R else []) for s in share]
        return [(j, per[j][i]) for i in range(max(len(p) for p in per)) for j in range(self.F) if i < len(per[j])]

    def prepare(self, *sample_counts):
        """Capture the remainder graphs the given run() sizes will need (outside any timed region)."""
        for n in sample_counts:
            for j, m in self._plan(n):
                self._slot(j, m)
        torch.cuda.synchronize()

    def _issue(self, j, n):
        s = self.slots[(j, n)]
        with torch.cuda.stream(self.streams[j]):
            if s[2] is not None:
                s[2].replay()
            else:
                s[1] = self._fwd(s[0])

    def run(self, n):
        for j, m in self._plan(n):
            self._issue(j, m)

    def verify(self):
        """Every sample of every slot equals the plain single-stream batch-1 forward of its voxel, bit for bit."""
        torch.cuda.synchronize()
        for (j, n), (x, o, g) in self.slots.items():
            for b in range(n):
                ref = self.model(x[b:b + 1])["flow"]
                assert all(torch.equal(a[b], r[0]) for a, r in zip(o["flow"], ref)), "an in-flight forward differs from the plain forward of its voxel"
            assert torch.isfinite(o["flow"][-1]).all()
        a = self.slots[(0, self.R)][1]["flow"][-1]
        if self.R > 1:
            assert not torch.equal(a[0], a[1]), "the samples of a launch sequence were meant to be different voxels"
        if self.F > 1:
            assert not torch.equal(a, self.slots[(1, self.R)][1]["flow"][-1]), "streams were meant to carry different voxels"

    def release(self):
        self.slots.clear()
        self._voxels.clear()


def inflight_rate(model, dev, F_, R, steps, warmup=6):
    """samples/s of `steps` batch-1 forwards in the headline's scheme (InFlight: F_ streams x R replicas per launch sequence; the timed
    region without the multi-rank plumbing), and the median latency of a synchronous forward."""
    with torch.no_grad():
        x0 = synthetic_chunk(1235).to(dev)
        for _ in range(3):
            model(x0)
        torch.cuda.synchronize()
        lat = []
        for _ in range(5):
            t0 = time.perf_counter()
            model(x0)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        fl = InFlight(model, dev, F_, R)
        fl.prepare(warmup, steps)
        fl.run(warmup)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fl.run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        fl.verify()
        fl.release()
    return {"samples_per_s": steps / dt, "ms_per_step": dt / steps * 1e3, "latency_ms_single_stream": sorted(lat)[len(lat) // 2],
            "steps": steps, "in_flight": F_, "replicas_per_launch": R}


def side_measurements(args, dev):
    """The other modes / BASELINE configurations beside the headline, each a short run of the same code paths (the default line's
    own configuration is never changed by them): the exact 3-plane and the 1-plane bf16 weight modes, the shipped PSN neuron,
    BASELINE configs[4] (20 bins / T = 20, 480 x 640, batch 4) and one configs[3] training step at local batch 4 on this GPU."""
    from sdformerflow_amd import train
    from sdformerflow_amd.harness import prepare_chunk
    from sdformerflow_amd.synthetic import synth_label, synth_voxel
    out = {}
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    # (the training step first: it allocates 25 GiB in its own pattern - behind the graph pools of the other side runs its first timed steps
    #  paid allocator growth: 109 ms in one run against 90.5 - 91.8 ms alone)
    try:
        m4, _ = build_model("lif", dev)
        m4.train()
        B = 4
        chunk = prepare_chunk(synth_voxel(B, 10, 288, 384, seed=1238)).to(dev)
        label, mask = (t.to(dev) for t in synth_label(B, 288, 384))
        buckets = train.GradientBuckets(m4.parameters())
        opt = torch.optim.AdamW(m4.parameters(), lr=1e-4, weight_decay=0.01)
        for _ in range(2):
            train.train_step(m4, opt, chunk, label, mask, buckets=buckets, dist=None, world=1, amp=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = [float(train.train_step(m4, opt, chunk, label, mask, buckets=buckets, dist=None, world=1, amp=False)) for _ in range(4)]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        assert all(v == v and abs(v) != float("inf") for v in losses)
        out["config4_train_step_1gpu"] = {"workload": "BASELINE configs[3] on ONE GPU: supervised training step (train-mode forward, loss, backward, clip, AdamW), "
                                                      "local batch 4, fp32, no collective at world size 1",
                                          "samples_per_s": B / dt, "ms_per_step": dt * 1e3, "loss_first_last": [losses[0], losses[-1]]}
        del m4, opt, buckets
    except Exception as e:
        out["config4_train_step_1gpu"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    try:                                                           # the headline's model in round 5's scheme: three batch-1 graphs in flight
        m, _ = build_model("lif", dev)
        r = inflight_rate(m, dev, 3, 1, 96)
        r["workload"] = "configs[1] forward, neuron=lif, weight planes=2: ONE sample per launch sequence, three streams (the scheme of rounds 1 - 5)"
        out["one_sample_per_launch_three_streams"] = r
        del m
    except Exception as e:
        out["one_sample_per_launch_three_streams"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    for name, kind, planes in (("planes3_exact_fp32_weights", "lif", 3), ("planes1_bf16_weights", "lif", 1), ("neuron_psn", "psn", 2)):
        try:
            m, _ = build_model(kind, dev)
            m.gemm_nsplit = planes
            # (the 16-bit-plane modes have no one-launch-sequence form - forward_replicas runs their samples one by one -: three streams)
            r = inflight_rate(m, dev, *((args.inflight, args.replicas) if planes == 2 else (3, 1)), 96)
            r["workload"] = f"configs[1] forward, neuron={kind}, weight planes={planes} ({ {1: 'one bf16', 2: 'fp16 hi+lo', 3: 'bf16 hi+mid+lo = fp32 exactly'}[planes] })"
            out[name] = r
            del m
        except Exception as e:                                     # a side line must never take the headline down with it
            out[name] = {"error": repr(e)[:300]}
        torch.cuda.empty_cache()
    # configs[4]: long-T / large-map stress, one stream (at batch 4 the launches fill the chip on their own)
    import yaml
    from sdformerflow_amd.STSwinNet_SNN.Spiking_STSwinNet import MS_SpikingformerFlowNet_en4
    from sdformerflow_amd.synthetic import synth_state_dict
    cfg = yaml.safe_load(open(os.path.join(ROOT, "sdformerflow_amd", "configs", "train_DSEC_supervised_SDformerFlow_en4.yml")))
    cfg["model"]["spiking_neuron"] = dict(cfg["spiking_neuron"], neuron_type="lif", num_steps=20)
    cfg["model"].update(num_bins=20)
    cfg["swin_transformer"].update(input_size=[480, 640])
    try:
        m5 = MS_SpikingformerFlowNet_en4(cfg["model"].copy(), cfg["swin_transformer"].copy())
        m5.load_state_dict(synth_state_dict({k: tuple(v.shape) for k, v in m5.state_dict().items()}), strict=True)
        m5 = m5.eval().to(dev)
        x5 = prepare_chunk(synth_voxel(4, 20, 480, 640, seed=1239)).to(dev)
        with torch.no_grad():
            for _ in range(2):
                o = m5(x5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                o = m5(x5)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 5
        assert torch.isfinite(o["flow"][-1]).all()
        out["config5_T20_480x640_batch4"] = {"workload": "BASELINE configs[4]: en4 forward, 20 bins / T = 20, 480x640, batch 4, neuron=lif, one stream (eager)",
                                             "samples_per_s": 4 / dt, "ms_per_batch": dt * 1e3}
        try:                                                       # two batches in flight (own stream, own HIP graph each): the headline's streams level
            x5b = prepare_chunk(synth_voxel(4, 20, 480, 640, seed=1249)).to(dev)
            with torch.no_grad():
                dt2, outs2 = streams_rate(lambda v: m5(v)["flow"], [x5, x5b], 8)
            assert all(torch.equal(a, b) for a, b in zip(outs2[0], o["flow"])), "a graph replay of configs[4] differs from the eager forward"
            out["config5_T20_480x640_batch4"]["two_batches_in_flight"] = {"samples_per_s": 4 / dt2, "ms_per_batch": dt2 * 1e3, "hip_graph": True}
            del outs2, x5b
        except Exception as e:
            out["config5_T20_480x640_batch4"]["two_batches_in_flight"] = {"error": repr(e)[:200]}
        del m5, x5, o
    except Exception as e:                                         # a side line must never take the headline down with it
        out["config5_T20_480x640_batch4"] = {"error": repr(e)[:300]}
    torch.cuda.empty_cache()
    torch.cuda.empty_cache()
    return out


def time_by_entry_point(model, chunk, top=5):
    """`roofline.by_time`: where one forward's kernel time goes.  One eager single-stream forward with every C-ABI call bracketed by two
    HIP events on its own stream (sdformerflow_amd.hip.profile_calls; a call is one kernel or a fixed short launch sequence, e.g. the
    attention half of a block), grouped by (entry point, shape): share of the summed call time, algorithmic work, achieved rate
    against the matching roof - the dense 16-bit MFMA peak for the matrix products, 8 TB/s for the byte-moving calls."""
    from sdformerflow_amd import hip
    with torch.no_grad():
        for _ in range(2):
            model(chunk)
        torch.cuda.synchronize()
        with hip.profile_calls() as rec:
            model(chunk)
    groups = {}
    for name, note, us in rec.rows():
        key = (name, tuple(note.get("shape", ())), bool(note.get("fused_neuron", False)))
        g = groups.setdefault(key, {"calls": 0, "us": 0.0, "flop": 0.0, "bytes": 0.0})
        g["calls"] += 1
        g["us"] += us
        g["flop"] += float(note.get("flop", 0))
        g["bytes"] += float(note.get("bytes", 0))
    total = sum(g["us"] for g in groups.values())
    rows = []
    for (name, shape, fused), g in sorted(groups.items(), key=lambda kv: -kv[1]["us"])[:top]:
        r = {"entry_point": name, "shape": list(shape), "calls_per_forward": g["calls"], "us_per_forward": g["us"], "share": g["us"] / total}
        if fused:
            r["fused_neuron_epilogue"] = True
        if g["flop"]:
            r.update(bound="mfma", algorithmic_gflop=g["flop"] / 1e9, tflops=g["flop"] / g["us"] / 1e6,
                     frac=g["flop"] / g["us"] / 1e6 / PEAK_BF16_DENSE_TFLOPS)
        elif g["bytes"]:
            r.update(bound="hbm", algorithmic_mbytes=g["bytes"] / 1e6, gbytes_per_s=g["bytes"] / g["us"] / 1e3, frac=g["bytes"] / g["us"] / 1e3 / 8000.0)
        rows.append(r)
    return {"rows": rows, "calls_per_forward": sum(g["calls"] for g in groups.values()), "sum_of_call_us": total,
            "note": "one eager single-stream forward, HIP events around every C-ABI call (events add ~2 us of bracket per call: shares, not "
                    "absolute launch times - those are in profiles/r4*_forward_sequence.txt)"}


KERNEL_COUNTERS = os.path.join(ROOT, "profiles", "r6_kernel_counters.json")      # tools/pmc_forward2.sh on the round's tree (per sample)
CLOCK_MHZ = 2400.0                                                               # peak engine clock (MI355X_MICROARCH.md)


def _short_kernel(n):
    import re
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("sdfmm::", "")
    return re.sub(r"\(.*", "", n)[:64]


def time_by_kernel(model, R, top=14):
    """`roofline.by_kernel`: one eager launch sequence of the headline's scheme (R samples, forward_replicas) through the library's own
    launch log (sdf_launch_log: HIP events around every kernel launch, grid and block noted) - per kernel, per SAMPLE: launches,
    duration, CHIP TIME = sum over its launches of min(workgroups / 256, 1) x duration (the share of the 256 compute units a launch can
    hold x the time it holds them: what adds up to the throughput when several launch sequences are in flight), and - from the
    counters of the same launch sequence taken by tools/pmc_forward2.sh on this tree (profiles/r6_kernel_counters.json; rocprofv3 owns
    the counters, they cannot be read here) - the kernel's VECTOR-ISSUE roof (wave-level VALU instructions x 4 cycles / 1024 SIMDs /
    2.4 GHz: the time its vector instructions alone need on the whole chip), matrix-pipe time (SQ_VALU_MFMA_BUSY_CYCLES / 1024 / 2.4
    GHz) and HBM time (bytes / 8 TB/s), each as a fraction of the measured duration: which resource a kernel is closest to."""
    from sdformerflow_amd import hip
    x = torch.cat([synthetic_chunk(1235 + i) for i in range(R)], 0).to(next(model.parameters()).device)
    fwd = (lambda: model.forward_replicas(x)) if R > 1 else (lambda: model(x))
    with torch.no_grad():
        for _ in range(2):
            fwd()
        torch.cuda.synchronize()
        with hip.launch_log() as log:
            fwd()
    agg = {}
    for k, wgs, thr, lds, us in log.rows:
        a = agg.setdefault(_short_kernel(k), {"launches": 0, "us": 0.0, "chip_us": 0.0, "max_workgroups": 0})
        a["launches"] += 1
        a["us"] += us
        a["chip_us"] += min(wgs / 256.0, 1.0) * us
        a["max_workgroups"] = max(a["max_workgroups"], wgs)
    counters, csrc = {}, None
    try:
        cj = json.load(open(KERNEL_COUNTERS))
        if cj.get("samples_per_launch_sequence") == R:
            counters, csrc = cj["per_sample"], "profiles/r6_kernel_counters.json"
    except (OSError, ValueError):
        pass
    tot_us, tot_chip = sum(a["us"] for a in agg.values()), sum(a["chip_us"] for a in agg.values())
    rows = []
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["chip_us"])[:top]:
        r = {"kernel": k, "launches": a["launches"], "us_per_sample": a["us"] / R, "chip_time_us_per_sample": a["chip_us"] / R,
             "share_of_chip_time": a["chip_us"] / tot_chip, "max_workgroups": a["max_workgroups"]}
        c = counters.get(k)
        if c:
            valu_us = c["valu_wave_insts"] * 4.0 / 1024.0 / CLOCK_MHZ
            mfma_us = c["mfma_busy_cycles"] / 1024.0 / CLOCK_MHZ
            hbm_us = (c["hbm_read_bytes"] + c["hbm_write_bytes"]) / (PEAK_HBM_GBPS * 1e3)
            d = a["us"] / R
            r.update(valu_issue_us=valu_us, valu_roof_frac=valu_us / d, matrix_pipe_us=mfma_us, matrix_pipe_frac=mfma_us / d, hbm_us=hbm_us,
                     hbm_frac=hbm_us / d, nearest_roof=max((("vector issue", valu_us), ("matrix pipe", mfma_us), ("hbm", hbm_us)), key=lambda t: t[1])[0])
        rows.append(r)
    return {"rows": rows, "launches_per_sequence": len(log.rows), "samples_per_sequence": R, "kernel_us_per_sample": tot_us / R,
            "chip_time_us_per_sample": tot_chip / R, "counters_from": csrc,
            "note": "chip time counts one workgroup per compute unit (an upper bound where several fit); events add ~1 us per launch"}


def time_swin_blocks(model, chunk, R=1, iters=10):
    """The attention-GEMM roofline fraction of the metric: SURVEY.md 8(d)'s 183.7 GFLOP of the swin blocks' Linear layers
    (q|k, proj, fc1, fc2, merge) per sample / the time of the swin stages themselves - the 12 blocks + 3 merges run alone on
    one stream between two HIP events (this includes the blocks' fused epilogues, gather neurons and token gates, so it is a
    lower bound of the GEMM launches' own rate; the per-launch split is in profiles/).  Measured twice: one sample per launch sequence
    (`swin_stages_ms`: the figure of rounds 2 - 5) and in the headline's scheme, R independent samples per launch sequence (replicas:
    `frac`, per sample) - the same kernels on R times the rows."""
    eng = model.engine()

    def run(y0, replicas):
        def stages(_):
            eng.replicas = replicas
            try:
                y = y0.clone()
                for s, blocks in enumerate(eng.stages):
                    for i in range(len(blocks)):
                        y = eng.swin_block(y, s, i)
                    if s < len(eng.merges):
                        y = eng.patch_merge(y, s)
            finally:
                eng.replicas = False
        return _timed(stages, [None], iters)
    with torch.no_grad():
        y0 = eng.patch_embed(chunk)
        t1 = run(y0, False)
        tR = run(torch.cat([y0] * R, 0).contiguous(), True) / R if R > 1 else t1
    return {"flop": 183.7e9, "swin_stages_ms": t1 * 1e3, "frac_one_sample_per_launch": 183.7e9 / t1 / (PEAK_BF16_DENSE_TFLOPS * 1e12),
            "replicas_per_launch": R, "swin_stages_ms_per_sample": tR * 1e3, "tflops": 183.7e9 / tR / 1e12,
            "frac": 183.7e9 / tR / (PEAK_BF16_DENSE_TFLOPS * 1e12),
            "note": "183.7 GFLOP (SURVEY.md 8d) per sample / time of the 12 swin blocks + 3 patch merges alone on one stream (HIP events), per "
                    "sample, in the headline's scheme (R samples per launch sequence; `frac_one_sample_per_launch` / `swin_stages_ms`: R = 1); "
                    "dense bf16 MFMA peak"}


def cpu_baseline(kind, sd, chunk, budget_s=25.0):
    """The oracle (a bit-level port of the reference's CPU path, see tests/test_oracle_golden.py) on the host cores."""
    from oracle import sdformer_oracle as O
    # cores = what this process may actually use (affinity mask and cgroup quota), not the host's core count:
    # oversubscribing a quota-limited container with one OpenMP thread per host core stalls for minutes
    cores = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    torch.set_num_threads(min(cores, torch.get_num_threads()))
    ocfg = {"neuron": O.NeuronCfg(kind, 0.1, None, 2.0, 10), "num_bins": 10, "window_size": (2, 9, 9),
            "depths": [2, 2, 6, 2], "num_heads": [3, 6, 12, 24]}
    sd = {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}
    with torch.no_grad():
        t0 = time.time()
        O.forward_flownet(chunk, sd, ocfg)                      # warm-up (also sizes the sample)
        first = time.time() - t0
        reps = max(0, min(5, int(budget_s / max(first, 1e-3)) - 1))
        t0 = time.time()
        for _ in range(reps):
            O.forward_flownet(chunk, sd, ocfg)
        dt = (time.time() - t0) / reps if reps else first
        reps = max(reps, 1)
    return {"value": 1.0 / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} full forwards of the same 1x10x2x288x384 voxel after 1 warm-up ({dt:.2f} s each), fp32, torch CPU"}


def max_over_ranks(dt, device, dist):
    """The timed region's length is the slowest rank's: MAX all-reduce of the per-rank wall time (the only data the
    inference path exchanges between ranks).  Also exercised on CPU with gloo (tests/test_replicas_gloo.py)."""
    if not dist:
        return dt
    import torch.distributed as td
    t = torch.tensor([dt], device=device, dtype=torch.float64)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(n_gpus, steps, dt):
    """Replicas: every rank runs `steps` forwards of its own sample in `dt` seconds -> aggregate samples / s."""
    return n_gpus * steps / dt


def main_train(args, world, rank, dev, dist, td):
    """BASELINE configs[3] (not the default workload): supervised training step of MS_SpikingformerFlowNet_en4 - train-mode
    forward, loss, backward, bucketed gradient all-reduce over RCCL (N > 1), clip, AdamW - on a local batch per GPU."""
    from sdformerflow_amd import train
    from sdformerflow_amd.harness import prepare_chunk
    from sdformerflow_amd.synthetic import synth_label, synth_voxel
    B = args.local_batch
    forward_fn, size = None, (288, 384)
    if args.plumbing:
        # CPU dry run of the N > 1 training plumbing (tests/test_bench_launcher.py, gloo): a few-layer torch stand-in with the model's
        # output contract (four flow maps) goes through the REAL train.train_step - grad-ready bucket hooks, the global valid-pixel
        # count, clip, AdamW - and the real rank records; no GPU work, and the line says so
        size = (32, 48)
        torch.manual_seed(7)                                     # every rank starts from the same weights, like a broadcast state_dict
        model = torch.nn.Sequential(torch.nn.Conv2d(20, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(16, 8, 3, padding=1)).to(dev)

        def forward_fn(m, c):
            y = m(c.reshape(c.shape[0], -1, *c.shape[-2:]))
            return [y[:, 2 * i:2 * i + 2] for i in range(4)]
    else:
        model, _ = build_model(args.neuron, dev)
    model.train()
    chunk = prepare_chunk(synth_voxel(B, 10, *size, seed=1238 + rank)).to(dev)
    label, mask = (t.to(dev) for t in synth_label(B, *size, seed=4321 + rank))
    buckets = train.GradientBuckets(model.parameters(), **({"bucket_bytes": 4096} if args.plumbing else {}))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)

    def barrier():
        if dist:
            td.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        train.train_step(model, opt, chunk, label, mask, buckets=buckets, dist=td if dist else None, world=world, amp=args.amp,
                         forward_fn=forward_fn)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(train.train_step(model, opt, chunk, label, mask, buckets=buckets, dist=td if dist else None, world=world,
                                       amp=args.amp, forward_fn=forward_fn))
    barrier()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, dev, dist)
    ranks = gather_ranks(rank_record(rank, dev, B * args.steps, dt_local), dist, td)
    losses = [float(v) for v in losses]
    assert all(v == v and abs(v) != float("inf") for v in losses), "non-finite loss"
    if rank == 0:
        check_ranks(ranks, world, shared_ok=os.environ.get("SDF_DIST_BACKEND", "nccl") != "nccl")
        n_gpus = world
        print(json.dumps({
            "metric": "training samples/sec (fwd+bwd+AdamW, 10-bin 288x384)" if not args.plumbing else
                      "training plumbing dry run (torch stand-in model on the CPU, not a benchmark result)",
            "value": n_gpus * B * args.steps / dt, "unit": "samples/s", "buckets": len(buckets.flat),
            "n_gpus": n_gpus, "world_size": world, "ranks": ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 autocast (fp32 neurons, BN statistics, loss, AdamW)" if args.amp else "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: MS_SpikingformerFlowNet_en4 supervised training step, local batch {B} per "
                                   f"GPU (global {n_gpus * B}), neuron={args.neuron}, AdamW 1e-4 / wd 0.01 / clip 100; spiking neurons "
                                   "forward + backward on HIP kernels, dense operators and their gradients on rocBLAS / MIOpen, "
                                   f"{len(buckets.flat)} flat gradient buckets all-reduced over RCCL", "local_batch": B,
                       "gradient_bytes": sum(b.numel() for b in buckets.flat) * 4},
            "loss_first_last": [losses[0], losses[-1]], "peak_memory_gib": torch.cuda.max_memory_allocated() / 2 ** 30 if dev.type == "cuda" else None}))
    if dist:
        td.barrier()
        td.destroy_process_group()


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def count_gpus_without_hip(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process could use, counted WITHOUT any HIP / HSA call (the launcher must never initialise the GPU: it starts
    the ranks as children): KFD topology nodes with SIMDs (CPU nodes have simd_count 0), narrowed by the *_VISIBLE_DEVICES
    lists when they are plain index lists.  None when the topology cannot be read - then the ranks verify (init_ranks)."""
    if not os.path.isdir("/sys/class/kfd") and topology.startswith("/sys/class/kfd"):
        return 0                                                 # no amdgpu compute driver at all: no GPU to give a rank
    try:
        nodes = sorted(os.listdir(topology), key=lambda v: int(v) if v.isdigit() else 1 << 30)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            props = dict(ln.split(None, 1) for ln in open(os.path.join(topology, node, "properties")) if " " in ln.strip())
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    if n == 0:
        return None                                              # unreadable / masked topology (containers): let the ranks verify
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = [t for t in v.split(",") if t.strip() != ""]
            if all(t.strip().isdigit() for t in ids):
                n = min(n, len(ids))
    return n


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process becomes the LAUNCHER.  It
    starts N ranks (one per GPU) through torch.distributed.run as a CHILD process and exits with its code; it never makes
    a GPU call itself - the device count comes from the KFD topology in sysfs, not from HIP (torch's own device count falls
    back to hipGetDeviceCount without amdsmi) - and never exec()s.  Every rank then proves its own GPU (init_ranks)."""
    import subprocess
    have = count_gpus_without_hip() if os.environ.get("SDF_DIST_BACKEND", "nccl") == "nccl" else None
    if have is not None and have < n:
        sys.stderr.write(f"bench.py: --gpus {n} but only {have} GPU(s) visible; refusing to report a "
                         f"{n}-GPU number from fewer devices\n")
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def init_ranks(args):
    """-> (world, rank, dev, dist, td).  The GPU count reported in the JSON is ALWAYS torch.distributed's world size (1 when
    not distributed), never the --gpus flag; the flag must agree with it, and every rank must own a distinct GPU (except with
    the explicit SDF_DIST_BACKEND=gloo test backend, where ranks may share a device or run the plumbing on the CPU)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SDF_DIST_BACKEND", "nccl")
    dist, td = world > 1, None
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
        # RCCL over xGMI (inference: timing barrier + max only; --train: the gradient all-reduce)
        td.init_process_group(backend)
        world = td.get_world_size()
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined (WORLD_SIZE); refusing to print a line")
    if args.plumbing:
        return world, rank, torch.device("cpu"), dist, td
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < world:
        raise SystemExit(f"bench.py: {world} ranks but {ndev} visible GPU(s): one GPU per rank is required")
    local_rank %= max(ndev, 1)                                   # only reachable with the gloo test backend
    torch.cuda.set_device(local_rank)
    return world, rank, torch.device("cuda", local_rank), dist, td


def gather_ranks(info, dist, td):
    """Every rank's {rank, device, ...} record on rank 0 (one all_gather_object; outside the timed region)."""
    if not dist:
        return [info]
    out = [None] * td.get_world_size()
    td.all_gather_object(out, info)
    return out


def rank_record(rank, dev, steps, dt_local):
    rec = {"rank": rank, "host_pid": os.getpid(), "device": str(dev), "samples_per_s": steps / dt_local}
    if dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        rec["device_name"] = pr.name
        rec["device_uuid"] = str(getattr(pr, "uuid", "")) or None
        rec["pci_bus_id"] = getattr(pr, "pci_bus_id", None)
    return rec


def check_ranks(ranks, world, shared_ok):
    """All `world` ranks reported, and (unless the gloo test backend shares devices on purpose) on distinct GPUs."""
    assert sorted(r["rank"] for r in ranks) == list(range(world)), f"ranks missing: {[r['rank'] for r in ranks]} of {world}"
    if not shared_ok:
        ids = [(r.get("device_uuid") or r["device"]) for r in ranks]
        assert len(set(ids)) == world, f"ranks share a GPU: {ids}"


def main_plumbing(args, world, rank, dist, td):
    """Launcher / rendezvous / barrier / max-over-ranks / rank-record plumbing without any GPU work (CPU test of the N > 1
    path, tests/test_bench_launcher.py).  The line it prints is labelled as such and is not a benchmark result."""
    dev = torch.device("cpu")
    if dist:
        td.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (rank + 1))
    if dist:
        td.barrier()
    dt_local = time.perf_counter() - t0
    dt = max_over_ranks(dt_local, dev, dist)
    ranks = gather_ranks(rank_record(rank, dev, args.steps, dt_local), dist, td)
    if rank == 0:
        check_ranks(ranks, world, shared_ok=True)
        print(json.dumps({"metric": "plumbing dry run (no GPU work, not a benchmark result)", "value": whole_job_rate(world, args.steps, dt),
                          "unit": "sleeps/s", "n_gpus": world, "world_size": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": dt / args.steps * 1e3, "data": "none", "ranks": ranks}))
    if dist:
        td.barrier()
        td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", action="store_true", help="BASELINE configs[3]: training step instead of the forward benchmark")
    ap.add_argument("--local-batch", type=int, default=4, help="--train: samples per GPU and step")
    ap.add_argument("--amp", action="store_true", help="--train: bf16 autocast for the dense operators (the reference uses fp16 amp)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 300 forwards / 10 training steps)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--neuron", default="lif", choices=["lif", "psn"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-config3", action="store_true", help="skip the BASELINE configs[2] (ANN, batch 8) side measurement")
    ap.add_argument("--no-sides", action="store_true", help="skip the other side measurements (weight-plane modes, PSN, configs[3] / [4])")
    ap.add_argument("--inflight", type=int, default=2, help="HIP streams per GPU, each replaying its own graph")
    ap.add_argument("--replicas", type=int, default=10, help="independent batch-1 forwards carried by ONE launch sequence (model.forward_replicas: "
                                                            "bit-equal to separate forwards); 1 = one sample per launch sequence")
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel instead of replaying HIP graphs")
    ap.add_argument("--planes", type=int, default=2, choices=[1, 2, 3],
                    help="16-bit weight planes of the spike GEMMs / convolutions: 2 = fp16 hi+lo (22 significand bits, the default "
                         "parity mode), 3 = bf16 hi+mid+lo (the fp32 weight exactly), 1 = one bf16 plane (BASELINE configs[1]'s "
                         "stated bf16 precision; a separately labelled throughput line, never the parity mode)")
    ap.add_argument("--plumbing", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--plan", default=None, help=argparse.SUPPRESS)       # diagnostic: the timed steps' dealing as "stream:samples,..."
    args = ap.parse_args()
    if args.inflight < 1 or args.replicas < 1:
        ap.error("--inflight and --replicas must be >= 1")
    if args.steps is None:
        args.steps = 10 if args.train else 300
    if args.warmup is None:
        args.warmup = 2 if args.train else 6
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    world, rank, dev, dist, td = init_ranks(args)
    if args.plumbing and not args.train:
        return main_plumbing(args, world, rank, dist, td)

    if args.train:
        return main_train(args, world, rank, dev, dist, td)

    model, sd = build_model(args.neuron, dev)
    model.gemm_nsplit = args.planes
    F_ = args.inflight
    # every (stream, position in the launch sequence) gets its OWN synthetic voxel (seed 1235 + k); voxel 0 is the one the CPU baseline runs
    chunk_cpu = synthetic_chunk(1235)
    chunk = chunk_cpu.to(dev)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # single-stream latency of one forward (reported beside the throughput; not part of the timed region)
        for _ in range(5):                                       # first calls: library algorithm search, allocator growth
            model(chunk)
        torch.cuda.synchronize()
        lat = []
        for _ in range(9):
            t0 = time.perf_counter()
            model(chunk)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        latency_ms = sorted(lat)[len(lat) // 2]                      # median of 9 synchronous forwards

        # F streams, each replaying its own HIP graph of one launch sequence over R independent samples (class InFlight)
        long_steps = max(args.steps, 96)
        fl = InFlight(model, dev, F_, args.replicas, eager=args.eager)
        if args.plan:
            fl.plan_override = [tuple(int(v) for v in p.split(":")) for p in args.plan.split(",")]
        fl.prepare(args.warmup, args.steps, long_steps)
        fl.run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        fl.run(args.steps)                                       # EXACTLY --steps forwards of one sample each
        barrier()
        dt_local = time.perf_counter() - t0
        # the same loop over >= 96 samples when the contract's --steps is shorter (a 20-step region is 30 ms: VERDICT r3); reported
        # beside `value`, which stays the contract's EXACTLY --steps
        dt_long = dt_local
        if long_steps > args.steps:
            barrier()
            t0 = time.perf_counter()
            fl.run(long_steps)
            barrier()
            dt_long = time.perf_counter() - t0
        # the sustained rate: the same loop for about two seconds (the chip's clock under a load of seconds is lower than in a burst
        # of a few hundred milliseconds: MI355X_MICROARCH.md, DVFS) - reported beside `value`, never instead of it
        n_sus = max(F_ * args.replicas, int(2.0 * long_steps / dt_long) // (F_ * args.replicas) * (F_ * args.replicas))
        barrier()
        t0 = time.perf_counter()
        fl.run(n_sus)
        barrier()
        dt_sus = time.perf_counter() - t0
        fl.verify()                                              # every in-flight sample == the plain forward of its voxel, bit for bit
    dt = max_over_ranks(dt_local, dev, dist)
    ranks = gather_ranks(rank_record(rank, dev, args.steps, dt_local), dist, td)

    if rank == 0:
        check_ranks(ranks, world, shared_ok=os.environ.get("SDF_DIST_BACKEND", "nccl") != "nccl")
        gemm, neuron = time_dominant_kernels(model, R=args.replicas)
        blocks = time_swin_blocks(model, chunk, args.replicas)
        by_time = time_by_entry_point(model, chunk)
        gemm["by_time"] = by_time["rows"]
        gemm["by_time_note"] = by_time["note"]
        gemm["calls_per_forward"] = by_time["calls_per_forward"]
        gemm["largest_share"] = by_time["rows"][0]["entry_point"] + " " + str(by_time["rows"][0]["shape"])
        try:
            gemm["by_kernel"] = time_by_kernel(model, args.replicas)
        except Exception as e:                                     # (diagnostic: never takes the line down)
            gemm["by_kernel"] = {"error": repr(e)[:300]}
        planes_txt = {1: "ONE bf16 weight plane (8 significand bits; throughput mode at BASELINE configs[1]'s stated precision, NOT the "
                         "parity mode)",
                      2: "2 fp16 weight planes hi+lo (22 of fp32's 24 significand bits; the default parity mode)",
                      3: "3 bf16 weight planes hi+mid+lo (the fp32 weights exactly)"}[args.planes]
        res = {
            "metric": "event-frames/sec fwd (1x10x2x288x384)", "value": whole_job_rate(world, args.steps, dt), "unit": "samples/s",
            "n_gpus": world, "world_size": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "value_over_90_steps": {"steps": long_steps, "samples_per_s_this_rank": long_steps / dt_long, "ms_per_step": dt_long / long_steps * 1e3},
            "value_sustained": {"steps": n_sus, "seconds": dt_sus, "samples_per_s_this_rank": n_sus / dt_sus, "ms_per_step": dt_sus / n_sus * 1e3,
                                "note": "the same timed loop run for ~2 s: the clock the chip holds under seconds of this load is lower than in the "
                                        "contract's short region"},
            "latency_ms_single_stream": latency_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {1: "bf16", 2: "f16x2", 3: "f32"}[args.planes], "data": "synthetic",
            "dtype_note": "binary u8 spikes x " + planes_txt + ", fp32 accumulate on the 16-bit MFMA pipe; membranes, BN, neurons fp32",
            "config": {"workload": "BASELINE configs[1]: MS_SpikingformerFlowNet_en4 forward, batch 1 per GPU, 10-bin 288x384 "
                                   "voxel, neuron=" + args.neuron + f"; a step = one forward of one sample; {args.inflight} HIP streams per GPU, each "
                                   f"{'launching' if args.eager else 'replaying the HIP graph of'} one launch sequence that carries {args.replicas} "
                                   "independent batch-1 forwards of different voxels (replicas: every flow map bit-equal to a separate batch-1 "
                                   "forward, asserted in this run); one process per GPU",
                       "in_flight": args.inflight, "replicas_per_launch": args.replicas, "samples_in_flight": args.inflight * args.replicas,
                       "hip_graph": not args.eager, "weight_planes": args.planes},
            "ranks": ranks,
            "roofline": gemm, "roofline_neuron": neuron,
            "attention_gemm_roofline_frac": blocks["frac"], "attention_gemm": blocks,
        }
        if not args.no_config3 and world == 1:
            res["config3_ann"] = time_config3()
        if not args.no_sides and world == 1 and args.planes == 2 and args.neuron == "lif":
            # the headline's graphs, static inputs and outputs are released first (their memory pools would stay beside the side runs),
            # and no side run can take the headline down: each is wrapped, the line is printed whatever happens
            fl.release()
            del model, fl
            torch.cuda.empty_cache()
            try:
                res["side_measurements"] = side_measurements(args, dev)
            except Exception as e:
                res["side_measurements"] = {"error": repr(e)[:300]}
        if not args.no_cpu and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.neuron, sd, chunk_cpu)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        # the numbers a reader needs again as the LAST key: the driver's record keeps the tail of this (long) line
        sm = res.get("side_measurements", {}) if isinstance(res.get("side_measurements"), dict) else {}
        res["headline_summary"] = {
            "value": res["value"], "unit": res["unit"], "steps": args.steps, "value_over_90_steps": res["value_over_90_steps"],
            "value_sustained_2s": res["value_sustained"]["samples_per_s_this_rank"],
            "latency_ms_single_stream": latency_ms, "roofline_frac": gemm.get("frac"), "attention_gemm_roofline_frac": blocks["frac"],
            "swin_stages_ms": blocks.get("swin_stages_ms"), "neuron_psn_samples_per_s": (sm.get("neuron_psn") or {}).get("samples_per_s"),
            "one_sample_per_launch_three_streams_samples_per_s": (sm.get("one_sample_per_launch_three_streams") or {}).get("samples_per_s"),
            "cpu_baseline_samples_per_s": (res.get("cpu_baseline") or {}).get("value")}
        print(json.dumps(res))
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
