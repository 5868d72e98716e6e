"""The one collective of the training path (BASELINE config 4): bucketed gradient all-reduce(sum)/world,
world_size 2 over gloo on CPU - the same `train.GradientBuckets.all_reduce` the GPU step calls over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, json, torch, torch.distributed as td
    sys.path.insert(0, os.environ["SDF_ROOT"])
    from sdformerflow_amd.train import GradientBuckets
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.Linear(200, 100), torch.nn.Linear(100, 3))
    dead = torch.nn.Parameter(torch.zeros(7))                      # a parameter that never receives a gradient (attn_sn)
    params = list(net.parameters()) + [dead]
    buckets = GradientBuckets(params, bucket_bytes=100 * 1024)     # forces several buckets
    buckets.zero()
    x = torch.full((4, 300), float(rank + 1))
    net(x).sum().backward()                                        # autograd accumulates into the bucket views
    local = [p.grad.clone() for p in params]
    buckets.all_reduce(td, world)
    # reference: gather every rank's local gradients and average
    ok = True
    for p, l in zip(params, local):
        parts = [torch.zeros_like(l) for _ in range(world)]
        td.all_gather(parts, l)
        ok = ok and torch.allclose(p.grad, sum(parts) / world, rtol=1e-6, atol=1e-7)
    same_views = all(p.grad.data_ptr() >= buckets.flat[i].data_ptr() for i, b in enumerate(buckets.buckets) for p in b)
    if rank == 0:
        print(json.dumps({"ok": bool(ok), "n_buckets": len(buckets.flat), "views": bool(same_views),
                          "dead_zero": bool((dead.grad == 0).all())}))
    td.barrier()
    td.destroy_process_group()
""")


def test_bucketed_gradient_all_reduce_two_ranks(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=240, cwd=ROOT, env=dict(os.environ, SDF_ROOT=ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["ok"] and r["views"] and r["dead_zero"] and r["n_buckets"] >= 2


WORKER_STEP = textwrap.dedent("""
    import os, sys, json, torch, torch.distributed as td
    sys.path.insert(0, os.environ["SDF_ROOT"])
    from sdformerflow_amd.train import GradientBuckets, flow_loss_supervised, global_valid_count
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()

    def make():
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.Tanh(), torch.nn.Conv2d(8, 8, 3, padding=1), torch.nn.Tanh(),
                                  torch.nn.Conv2d(8, 2, 3, padding=1))
        dead = torch.nn.Parameter(torch.ones(5))                   # never receives a gradient (the PSN model's attn_sn)
        return net, dead

    g = torch.Generator().manual_seed(1)
    X = torch.randn(4, 3, 12, 16, generator=g)                     # the GLOBAL batch: rank r owns samples 2r, 2r+1
    Y = torch.randn(4, 2, 12, 16, generator=g) * 3
    M = (torch.rand(4, 12, 16, generator=g) < 0.7 - 0.4 * torch.arange(4).view(4, 1, 1) / 4).float()   # valid counts differ per sample

    # --- single process, whole batch: the reference's gathered-batch loss (DataParallel)
    net, dead = make()
    flow_loss_supervised([net(X)], Y, M).backward()
    want = [p.grad.clone() for p in net.parameters()]

    # --- this rank's shard through GradientBuckets with the hooks + global valid count
    net, dead = make()
    params = list(net.parameters()) + [dead]
    buckets = GradientBuckets(params, bucket_bytes=1024)           # several buckets
    out = {}
    for step in range(2):                                          # second step: the dead parameter's view must come back / go again
        buckets.begin(td, world)
        sl = slice(2 * rank, 2 * rank + 2)
        n = global_valid_count(M[sl], td, world)
        loss = flow_loss_supervised([net(X[sl])], Y[sl], M[sl], n_valid=n)
        loss.backward()
        log_before_finish = list(buckets.log)
        buckets.finish()
        got = [p.grad for p in net.parameters()]
        out[step] = {"match": all(torch.allclose(a, b, rtol=1e-5, atol=1e-7) for a, b in zip(got, want)),
                     "max_rel": max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(got, want)),
                     "dead_none": dead.grad is None, "n_global": float(n), "n_expected": float(M.sum()),
                     "launched_in_backward": len(log_before_finish), "n_buckets": len(buckets.flat),
                     "first_launch_hooks": log_before_finish[0][2] if log_before_finish else None,
                     "n_live": len(list(net.parameters())), "views": all(p.grad.data_ptr() == buckets._views[p].data_ptr() for p in net.parameters())}
    # the local-count normalisation the advisor flagged: ~world x the gathered-batch gradient
    net2, _ = make()
    flow_loss_supervised([net2(X[sl])], Y[sl], M[sl]).backward()
    ratio = float(torch.cat([p.grad.flatten() for p in net2.parameters()]).norm() / torch.cat([w.flatten() for w in want]).norm())
    orders = [None] * world
    td.all_gather_object(orders, [e[1] for e in buckets.log])
    if rank == 0:
        print(json.dumps({"steps": out, "same_order_on_all_ranks": all(o == orders[0] for o in orders), "local_norm_ratio": ratio}))
    td.barrier()
    td.destroy_process_group()
""")


def test_overlapped_buckets_reproduce_the_gathered_batch_gradient(tmp_path):
    """Two ranks, each with half of a 4-sample batch: gradients after the hook-launched bucket all-reduces equal the
    single-process gradient of the reference's gathered-batch loss (global valid-pixel count); the first bucket's
    collective is launched INSIDE backward, before the remaining parameters have their gradients; a parameter without a
    gradient reaches the optimiser as None; every rank launches the buckets in the same order."""
    script = tmp_path / "w2.py"
    script.write_text(WORKER_STEP)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=240, cwd=ROOT, env=dict(os.environ, SDF_ROOT=ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    for st in r["steps"].values():
        assert st["match"], st
        assert st["dead_none"] and st["views"]
        assert st["n_global"] == st["n_expected"]
        assert st["n_buckets"] >= 3
        assert st["launched_in_backward"] >= st["n_buckets"] - 1         # all but the bucket holding the dead parameter
        assert st["first_launch_hooks"] < st["n_live"]                    # launched before backward had finished
    assert r["same_order_on_all_ranks"]
    assert r["local_norm_ratio"] > 1.3                                     # what local normalisation would have given (~world x)
