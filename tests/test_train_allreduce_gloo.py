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
