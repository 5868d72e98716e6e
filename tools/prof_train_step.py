#!/usr/bin/env python3
"""Kernel-time breakdown of ONE steady-state training step from a rocprofv3 --kernel-trace of `bench.py --train`
(the whole-run --stats table is dominated by MIOpen's one-time algorithm search).  usage: prof_train_step.py <dir> <ms per step>"""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
win = float(sys.argv[2]) * 1e6
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
end = int(rows[-1]["End_Timestamp"])
sel = [r for r in rows if int(r["Start_Timestamp"]) > end - win]
groups = (("linear_dw_kernel", "ours: Linear / conv weight gradient"), ("linear_dw_reduce", "ours: weight-gradient range sums"), ("ringed_rows", "ours: ringed channels-last rows"),
          ("lif_bwd_kernel", "ours: LIF backward"), ("neuron_kernel", "ours: neuron forward"), ("psn_bwd", "ours: PSN backward"),
          ("bn_reduce", "ours: batch-norm reductions"), ("bn_apply", "ours: batch-norm apply"), ("bn_finish", "ours: batch-norm finish"),
          ("qk_gate_train", "ours: token gate fwd / bwd"), ("Cijk", "rocBLAS GEMM"), ("miopenSp3AsmConv", "MIOpen Winograd conv"),
          ("igemm", "MIOpen implicit-GEMM conv"), ("BatchNorm", "MIOpen batch-norm (NCHW conv outputs)"),
          ("direct_copy", "ATen layout copies"), ("CUDAFunctor_add", "ATen adds"), ("transpose", "MIOpen transposes"),
          ("multi_tensor", "AdamW / clip"), ("reduce_kernel", "ATen reductions"), ("roll", "ATen roll"), ("elementwise", "ATen other elementwise"))
acc = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r["Kernel_Name"]
    label = next((lab for key, lab in groups if key in n), n[:50])
    acc[label][0] += 1
    acc[label][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in acc.values())
print(f"last {win/1e6:.0f} ms of the trace: {len(sel)} launches, kernel time {tot:.1f} ms")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:26]:
    print(f"{v[1]:8.2f} ms  x{v[0]:<5d} {k}")

# the library launches one by one (what is left to replace): duration, grid
lib = [r for r in sel if any(k in r["Kernel_Name"] for k in ("igemm", "miopenSp3AsmConv", "Cijk"))]
print("library convolution / GEMM launches of the step, largest first:")
for r in sorted(lib, key=lambda r: int(r["Start_Timestamp"]) - int(r["End_Timestamp"]))[:44]:
    print(f"  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', '?'):>8s}  {r['Kernel_Name'][:110]}")
