#!/usr/bin/env bash
# HBM bytes of ONE forward of BASELINE config 2: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, summed over the kernels of the
# last of three eager forwards (MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950), with the per-kernel-name breakdown.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
rm -rf gpurun_out/pmc_forward
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_forward/$set -- python3 tools/forward_one.py > /dev/null 2>&1 < /dev/null
done
python3 - <<PY
import csv, glob, collections
tot = {}
per = collections.defaultdict(lambda: [0.0, 0.0, 0])
for ci, name in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    fs = glob.glob(f"gpurun_out/pmc_forward/{name}/**/*counter_collection.csv", recursive=True)
    rows = [r for f in fs for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    heads = [i for i, r in enumerate(rows) if "head_conv_sn_kernel" in r["Kernel_Name"]]      # once per forward, near its start
    last = rows[heads[-1] - (heads[-1] - heads[-2] - (len(rows) - heads[-1])):] if len(heads) > 1 else rows
    # = the last forward: from as many launches before its head convolution as the previous forward had after its own tail
    tot[name] = sum(float(r["Counter_Value"]) for r in last)
    for r in last:
        k = r["Kernel_Name"][:90]
        per[k][ci] += float(r["Counter_Value"]); per[k][2] += 1 if ci == 0 else 0
rd, wr = 2 * tot["FETCH_SIZE"] / 1e6, tot["WRITE_SIZE"] / 1e6
print(f"one forward of config 2 (eager, one stream): HBM read {rd:.2f} GB (FETCH_SIZE x 2) + written {wr:.2f} GB = {rd + wr:.2f} GB")
for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1][0] + kv[1][1]))[:16]:
    print(f"  read {2 * v[0] / 1e3:8.1f} MB  written {v[1] / 1e3:8.1f} MB  x{v[2]:3d}  {k}")
PY
