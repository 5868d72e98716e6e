"""N>1 path of bench.py's protocol on CPU: world_size-2 gloo, barrier + MAX all-reduce of the step time, rank 0
aggregates `value = N*K/max(t)` - the only collective the inference path has (replicas)."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time, json, torch, torch.distributed as td
    sys.path.insert(0, os.environ["SDF_ROOT"])
    import bench                                  # the functions bench.py's main() itself uses for N > 1
    td.init_process_group("gloo")
    rank, world = td.get_rank(), td.get_world_size()
    td.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))            # ranks finish at different times
    td.barrier()
    dt = time.perf_counter() - t0
    mx = bench.max_over_ranks(dt if rank else dt * 0.5, torch.device("cpu"), True)   # rank 0 pretends to be faster
    if rank == 0:
        print(json.dumps({"n": world, "max_t": mx, "mine": dt, "rate": bench.whole_job_rate(world, 20, mx)}))
    td.barrier()
    td.destroy_process_group()
""")


def test_two_rank_gloo_timing_protocol(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=180, cwd=ROOT, env=dict(os.environ, SDF_ROOT=ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n"] == 2 and r["max_t"] >= 0.1 - 1e-3            # the slower rank (0.10 s) defines the step time
    assert r["max_t"] > 0.6 * r["mine"]                         # rank 0 reported half its time: the maximum came from rank 1
    assert abs(r["rate"] - 2 * 20 / r["max_t"]) < 1e-6          # whole-job rate = all ranks' samples / slowest rank's time
