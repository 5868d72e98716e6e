"""`python bench.py --gpus N` must START N ranks itself when it is not already running under torch.distributed.run, and the
GPU count in its JSON line must be the number of ranks that joined - never the flag.  Run here end to end on the CPU with
the explicit gloo test backend and the `--plumbing` dry run (launcher, rendezvous, barrier, max-over-ranks and the per-rank
records are bench.py's own code paths; no GPU work, and the line says so)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(SDF_DIST_BACKEND="gloo", **(env_extra or {}))
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, (out.stdout[-2000:], out.stderr[-2000:])       # ONE JSON line, from rank 0
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_reports_them():
    out = _run(["--gpus", "2", "--steps", "20", "--warmup", "1", "--plumbing"])
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["world_size"] == 2
    assert sorted(x["rank"] for x in r["ranks"]) == [0, 1]
    assert len({x["host_pid"] for x in r["ranks"]}) == 2                     # two real processes
    # whole-job value = all ranks' steps / the SLOWEST rank's time (rank 1 sleeps twice as long per step)
    slow = min(x["samples_per_s"] for x in r["ranks"])
    assert abs(r["value"] - 2 * 20 / (r["ms_per_step"] * 20e-3)) < 1e-6 * r["value"]
    assert r["value"] <= 2.0 * slow * 1.05
    assert r["value"] < sum(x["samples_per_s"] for x in r["ranks"])           # not N x the fastest rank


def test_single_process_line_reports_one_rank():
    r = _line(_run(["--steps", "5", "--warmup", "0", "--plumbing"]))
    assert r["n_gpus"] == 1 and r["world_size"] == 1 and [x["rank"] for x in r["ranks"]] == [0]


def test_flag_that_disagrees_with_the_joined_ranks_is_refused():
    """Started as ONE process that is told it is rank 0 of a world of 1 while --gpus says 8: no line, non-zero exit
    (round 1 printed 8 x the single-GPU rate here)."""
    out = _run(["--gpus", "8", "--steps", "5", "--plumbing"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"},
               drop=("MASTER_ADDR", "MASTER_PORT"))
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_launcher_refuses_more_ranks_than_gpus_on_the_real_backend():
    """Without the gloo test backend the launcher checks the visible device count BEFORE starting anything (this container
    has no GPU): exit 2, no line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SDF_DIST_BACKEND")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    assert out.returncode == 2 and "refusing" in out.stderr


def test_launcher_counts_gpus_from_the_kfd_topology_without_hip(tmp_path, monkeypatch):
    """The device count the launcher refuses on comes from sysfs (KFD topology nodes with SIMDs), never from a HIP call."""
    sys.path.insert(0, ROOT)
    import bench
    for i, simd in enumerate([0, 0, 256, 256, 256]):                          # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {8 if simd == 0 else 0}\nsimd_count {simd}\ngfx_target_version {90500 if simd else 0}\n")
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.count_gpus_without_hip(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.count_gpus_without_hip(str(tmp_path)) == 2
    assert bench.count_gpus_without_hip(str(tmp_path / "missing")) is None        # unreadable: the ranks verify themselves
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks"):src.index("def init_ranks")]
    assert "torch.cuda" not in body                                               # the launcher makes no torch.cuda call at all


def test_train_gpus_2_runs_the_real_step_under_gloo():
    """`bench.py --train --gpus 2` end to end on the CPU: launcher, rendezvous, then the REAL
    sdformerflow_amd.train.train_step on a torch stand-in model - GradientBuckets with several buckets whose all-reduces leave from the
    grad-ready hooks, the global valid-pixel count (each rank has its own label / mask), clip, AdamW - and the rank records.  The two
    ranks see different data, so a loss that falls on both and equal parameters at the end mean the gradients were averaged."""
    out = _run(["--train", "--gpus", "2", "--steps", "6", "--warmup", "1", "--local-batch", "2", "--plumbing"])
    assert out.returncode == 0, out.stderr[-3000:]
    r = _line(out)
    assert r["n_gpus"] == 2 and r["world_size"] == 2 and sorted(x["rank"] for x in r["ranks"]) == [0, 1]
    assert len({x["host_pid"] for x in r["ranks"]}) == 2
    assert r["buckets"] >= 2                                                  # the hook-launched path with more than one bucket
    assert "dry run" in r["metric"] and r["config"]["local_batch"] == 2
    first, last = r["loss_first_last"]
    assert last < first and last == last
    assert abs(r["value"] - 2 * 2 * 6 / (r["ms_per_step"] * 6e-3)) < 1e-6 * r["value"]


def test_inflight_plan_deals_exactly_n_samples_evenly():
    """bench.InFlight._plan: the K timed steps (one forward of one sample each) are dealt evenly to the F streams, a stream's share as
    replays of its R-sample graph plus one remainder graph - EXACTLY K samples whatever R and F are (the contract's "time exactly K steps")."""
    import bench
    for R in (1, 3, 4, 10):
        for F_ in (1, 2, 3):
            fl = bench.InFlight.__new__(bench.InFlight)
            fl.F, fl.R = F_, R
            for n in (1, 5, 20, 96, 97, 300):
                plan = fl._plan(n)
                assert sum(m for _, m in plan) == n and all(0 < m <= R and 0 <= j < F_ for j, m in plan)
                share = [sum(m for j, m in plan if j == s) for s in range(F_)]
                assert max(share) - min(share) <= 1                      # evenly over the streams
                assert all(sum(1 for j, m in plan if j == s and m < R) <= 1 for s in range(F_))     # at most one remainder graph per stream
