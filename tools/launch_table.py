#!/usr/bin/env python3
"""Per-launch table of ONE forward of the headline configuration from the library's own launch log (hip.launch_log: HIP events around
every kernel launch, no profiler): duration, workgroups, chip time = min(workgroups / resident slots, 1) x duration, per launch and
summed per kernel.   usage: launch_table.py [lif|psn] [R = 1] [seq]      R > 1: forward_replicas over R samples; seq: also the sequence"""
import os, re, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from sdformerflow_amd import hip
kind = sys.argv[1] if len(sys.argv) > 1 else "lif"
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
model, _ = bench.build_model(kind, dev)
x = torch.cat([bench.synthetic_chunk(1235 + i) for i in range(R)], 0).to(dev)
fwd = (lambda: model.forward_replicas(x)) if R > 1 else (lambda: model(x))
with torch.no_grad():
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    with hip.launch_log() as log:
        fwd()


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\b(sdfmm|sdf)::", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def chip_us(wgs, us):
    return min(wgs / 256.0, 1.0) * us               # (one workgroup per compute unit as the unit: an upper bound where several fit a CU)


tot = sum(r[4] for r in log.rows)
chip = sum(chip_us(r[1], r[4]) for r in log.rows)
print(f"# {kind}, {R} sample(s) per launch sequence: {len(log.rows)} launches, {tot:.1f} us of kernel time = {tot / R:.1f} us per sample; "
      f"chip time {chip:.1f} us = {chip / R:.1f} us per sample")
if "seq" in sys.argv:
    for i, (k, wgs, thr, lds, us) in enumerate(log.rows):
        print(f"{i:4d} {us:8.1f} us {wgs:6d} wg x {thr:4d} thr {lds:6d} B lds  chip {chip_us(wgs, us):7.1f}  {short(k)}")
agg = {}
for k, wgs, thr, lds, us in log.rows:
    a = agg.setdefault(short(k), [0, 0.0, 0.0, 0])
    a[0] += 1; a[1] += us; a[2] += chip_us(wgs, us); a[3] = max(a[3], wgs)
print(f"{'kernel':72s} {'n':>4s} {'us':>9s} {'us/sample':>10s} {'chip us':>9s} {'max wgs':>8s}")
for k, (n, us, ch, mw) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:72s} {n:4d} {us:9.1f} {us / R:10.1f} {ch:9.1f} {mw:8d}")
