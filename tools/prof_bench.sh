#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the one the driver runs) -> gpurun_out/prof_$1
set -e
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/bench.py --no-cpu --no-sides > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
tail -c 600 gpurun_out/prof_$TAG.log
python3 - <<PY | tee gpurun_out/prof_${TAG}_kernel_stats.txt
import csv, glob
f = glob.glob("gpurun_out/prof_$TAG/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("rocprofv3 --kernel-trace --stats of: python3 bench.py --no-cpu --no-sides   (the default scheme: 300 steps, 10 samples per launch sequence, 2 streams, HIP-graph replay; "
      "the roofline / by_kernel / config-3 measurements of the same command are in the table too) - EVERY kernel of the run:")
for r in rows:
    print(f"{r['Name'][:110]:110s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
