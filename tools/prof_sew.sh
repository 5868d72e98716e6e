#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of tools/sew_try.py: which launches of a SEW forward are library kernels (rocBLAS / MIOpen / ATen)
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sew
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_sew -o sew -- python3 $R/tools/sew_try.py lif 10 > $R/gpurun_out/prof_sew.log 2>&1
cd $R
tail -2 gpurun_out/prof_sew.log
python3 - <<'PY' | tee gpurun_out/prof_sew_stats.txt
import csv, glob
f = glob.glob("gpurun_out/prof_sew/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time of 13 forwards + engine build: {tot / 1e6:.1f} ms")
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs']) / 1e3 / 13:9.1f} us/fwd  x{int(r['Calls']) / 13:6.1f}  {100 * float(r['TotalDurationNs']) / tot:5.1f} %  {r['Name'][:110]}")
PY
rm -rf gpurun_out/prof_sew
