#!/usr/bin/env python3
"""Benchmark of the SDformerFlow forward hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--neuron lif|psn] [--no-cpu] [--inflight F] [--eager]
    python bench.py --train [--local-batch B] ...      (BASELINE configs[3], the training step; not the default workload)

One "step" = one forward of MS_SpikingformerFlowNet_en4 over one synthetic 1 x 10 x 2 x 288 x 384 event
voxel (BASELINE config 2) already resident in HBM.  For N > 1 the driver launches one process per GPU
(torch.distributed.run); the forward does not shard inside a micro-batch (batch elements are
coupled by the reference's raw reshapes), so every rank runs an independent replica: weak scaling, no
data-path collective, value = N*K / max-over-ranks(time).

Every rank keeps F (default 3) independent forwards in flight on F HIP streams, each captured once as a HIP graph and
replayed (`--eager` launches kernel by kernel instead): the layers of a batch-1 forward are too small to fill 256 CUs one
kernel at a time, independent samples overlap.  A step is still one whole forward of one sample; the K timed steps are dealt
round-robin to the streams.  The single-stream latency is reported next to the throughput.

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
CONV_TRAFFIC_BYTES = (2 * 75516.1 + 104192.0) * 1024     # PMC, see roofline.traffic_source


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


def synthetic_chunk():
    """Harness input prep of eval_DSEC_flow_SNN.py:179-205 on a seeded DSEC-like voxel."""
    from sdformerflow_amd.harness import prepare_chunk
    from sdformerflow_amd.synthetic import synth_voxel
    return prepare_chunk(synth_voxel(1, 10, 288, 384, seed=1235))


def time_dominant_kernels(model, iters=20):
    """Live HIP-event timing (on the launch stream = torch's current stream, the one passed through the C ABI) of the
    dominant kernel of the forward - the warp-specialised spike convolution on the patch-embedding res-block shape
    (10 images of 144x192, 96 -> 96 channels, 3x3: 4 of the 26 convolution launches and ~0.8 ms of the step) - and of
    the HBM-bound neuron update."""
    from sdformerflow_amd import hip
    eng = model.engine()
    dev = eng.device
    imgs, H, W, Cc = 10, 144, 192, 96
    rb = eng.pe_res[0]
    x = (torch.rand((imgs, H, W, Cc), device=dev) < 0.3).to(torch.uint8)
    out = torch.empty((imgs * H * W, Cc), device=dev)
    resid = torch.rand((imgs * H * W, Cc), device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def conv():
        hip.spike_conv2d(x, rb.w2, imgs, H, W, Cc, H, W, 3, 3, 1, (-1, 0, 1), (-1, 0, 1), out=out, alpha=rb.bn2[0],
                         beta=rb.bn2[1], resid=resid)
    for _ in range(3):
        conv()
    e0.record()
    for _ in range(iters):
        conv()
    e1.record()
    torch.cuda.synchronize()
    t_conv = e0.elapsed_time(e1) / iters * 1e-3
    flops = 2.0 * imgs * H * W * Cc * 9 * Cc                    # algorithmic: one multiply-add per (pixel, cout, tap, cin)
    ns = int(rb.w2.shape[0])
    gemm = {"kernel": f"sdfmm::spike_mm_pp_kernel<{ns},0,true> (3x3 spike conv 96->96 @ 10x144x192, BN + residual epilogue)",
            "bound": "mfma", "achieved": flops / t_conv / 1e12, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s",
            "us_per_launch": t_conv * 1e6, "traffic": CONV_TRAFFIC_BYTES if ns == 2 else None,
            "traffic_unit": "bytes per launch (HBM read + write)",
            "traffic_source": "profiles/r1m_pmc_traffic_conv.txt: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) and WRITE_SIZE in separate "
                              "passes on this exact launch; algorithmic bytes 239.2e6",
            "note": f"algorithmic flops (2 per multiply-add of the convolution); the kernel issues {ns} 16-bit MFMAs per product "
                    f"(fp32 weights carried as {ns} planes), i.e. {ns}x this on the matrix pipe; dense peak of the f16/bf16 MFMA"}
    gemm["frac"] = gemm["achieved"] / gemm["peak"]
    # neuron: T=10 over the stage-0 MLP hidden tensor shape (10 x 72*96*384 fp32 in, u8 out)
    blk = eng.stages[0][0]
    n = 72 * 96 * 384
    xx = torch.rand((10, n), device=dev) - 0.3
    s = torch.empty((10, n), dtype=torch.uint8, device=dev)
    p = blk.sn2
    for _ in range(3):
        hip.neuron_fwd(xx, s, 10, 1, n, 0, n, 0, n, p)
    e0.record()
    for _ in range(iters):
        hip.neuron_fwd(xx, s, 10, 1, n, 0, n, 0, n, p)
    e1.record()
    torch.cuda.synchronize()
    t_n = e0.elapsed_time(e1) / iters * 1e-3
    neuron = {"kernel": "neuron_kernel<10> (26.5 M neurons x T=10, f32 in / u8 out)", "bound": "hbm",
              "achieved": 10.0 * n * 5 / t_n / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "us_per_launch": t_n * 1e6,
              "traffic": None}
    neuron["frac"] = neuron["achieved"] / neuron["peak"]
    return gemm, neuron


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
    model, _ = build_model(args.neuron, dev)
    model.train()
    B = args.local_batch
    chunk = prepare_chunk(synth_voxel(B, 10, 288, 384, seed=1238 + rank)).to(dev)
    label, mask = (t.to(dev) for t in synth_label(B, 288, 384))
    buckets = train.GradientBuckets(model.parameters())
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.01)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize()

    losses = []
    for _ in range(args.warmup):
        train.train_step(model, opt, chunk, label, mask, buckets=buckets, dist=td if dist else None, world=world, amp=args.amp)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(train.train_step(model, opt, chunk, label, mask, buckets=buckets, dist=td if dist else None, world=world,
                                       amp=args.amp))
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, dev, dist)
    losses = [float(v) for v in losses]
    assert all(v == v and abs(v) != float("inf") for v in losses), "non-finite loss"
    if rank == 0:
        n_gpus = world if dist else args.gpus
        print(json.dumps({
            "metric": "training samples/sec (fwd+bwd+AdamW, 10-bin 288x384)", "value": n_gpus * B * args.steps / dt, "unit": "samples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16 autocast (fp32 neurons, BN statistics, loss, AdamW)" if args.amp else "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[3]: MS_SpikingformerFlowNet_en4 supervised training step, local batch {B} per "
                                   f"GPU (global {n_gpus * B}), neuron={args.neuron}, AdamW 1e-4 / wd 0.01 / clip 100; spiking neurons "
                                   "forward + backward on HIP kernels, dense operators and their gradients on rocBLAS / MIOpen, "
                                   f"{len(buckets.flat)} flat gradient buckets all-reduced over RCCL", "local_batch": B,
                       "gradient_bytes": sum(b.numel() for b in buckets.flat) * 4},
            "loss_first_last": [losses[0], losses[-1]], "peak_memory_gib": torch.cuda.max_memory_allocated() / 2 ** 30}))
    if dist:
        td.barrier()
        td.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train", action="store_true", help="BASELINE configs[3]: training step instead of the forward benchmark")
    ap.add_argument("--local-batch", type=int, default=4, help="--train: samples per GPU and step")
    ap.add_argument("--amp", action="store_true", help="--train: bf16 autocast for the dense operators (the reference uses fp16 amp)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--neuron", default="lif", choices=["lif", "psn"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--inflight", type=int, default=3, help="independent forwards in flight per GPU (HIP streams)")
    ap.add_argument("--eager", action="store_true", help="launch kernel by kernel instead of replaying HIP graphs")
    args = ap.parse_args()
    if args.inflight < 1:
        ap.error("--inflight must be >= 1")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = world > 1
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
        # RCCL over xGMI (inference: timing barrier only; --train: the gradient all-reduce).  SDF_DIST_BACKEND=gloo lets the
        # N > 1 code path be exercised with several ranks sharing one GPU, which RCCL refuses
        td.init_process_group(os.environ.get("SDF_DIST_BACKEND", "nccl"))
    ndev = max(torch.cuda.device_count(), 1)
    local_rank %= ndev                                           # more ranks than visible GPUs: share (never silently fail)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.train:
        return main_train(args, world, rank, dev, dist, td if dist else None)

    model, sd = build_model(args.neuron, dev)
    chunk_cpu = synthetic_chunk()
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

        # F independent forwards in flight: own stream, own static input, own graph (own activation memory)
        F_ = args.inflight
        streams = [torch.cuda.Stream(device=dev) for _ in range(F_)]
        inputs, outputs, graphs = [], [], []
        for st in streams:
            x = chunk.clone()
            with torch.cuda.stream(st):
                for _ in range(2):
                    o = model(x)                                 # also creates this stream's split-K workspace and plan caches
            torch.cuda.synchronize()
            g = None
            if not args.eager:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    o = model(x)
            inputs.append(x); outputs.append(o); graphs.append(g)
        torch.cuda.synchronize()

        def step(i):
            j = i % F_
            with torch.cuda.stream(streams[j]):
                if graphs[j] is not None:
                    graphs[j].replay()
                else:
                    outputs[j] = model(inputs[j])

        for i in range(args.warmup):
            step(i)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        barrier()
        dt = time.perf_counter() - t0
    out = outputs[0]
    with torch.no_grad():
        ref = model(chunk)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for o in outputs for a, b in zip(o["flow"], ref["flow"])), "in-flight forwards differ from a plain one"
    assert torch.isfinite(out["flow"][-1]).all()
    dt = max_over_ranks(dt, dev, dist)

    if rank == 0:
        n_gpus = world if dist else args.gpus
        gemm, neuron = time_dominant_kernels(model)
        res = {
            "metric": "event-frames/sec fwd (1x10x2x288x384)", "value": whole_job_rate(n_gpus, args.steps, dt), "unit": "samples/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "latency_ms_single_stream": latency_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: MS_SpikingformerFlowNet_en4 forward, batch 1 per GPU, 10-bin 288x384 "
                                   "voxel, neuron=" + args.neuron + f"; spike GEMMs and spike convolutions on 16-bit MFMA with "
                                   f"{model.gemm_nsplit}-plane fp32-grade weights and fp32 accumulate; {args.inflight} independent forwards in flight "
                                   f"per GPU on HIP streams ({'eager launches' if args.eager else 'HIP-graph replay'}); replicas per GPU",
                       "in_flight": args.inflight, "hip_graph": not args.eager},
            "roofline": gemm, "roofline_neuron": neuron,
            "attention_gemm_roofline_frac": 183.7e9 / (dt / args.steps) / (PEAK_BF16_DENSE_TFLOPS * 1e12),
        }
        if not args.no_cpu and n_gpus == 1:
            res["cpu_baseline"] = cpu_baseline(args.neuron, sd, chunk_cpu)
            res["gpu_over_cpu"] = res["value"] / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
