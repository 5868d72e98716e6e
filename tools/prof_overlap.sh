#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of the default bench command (no side measurements) + the overlap report
set -e
TAG=${1:-r3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/bench.py --no-cpu --no-sides --no-config3 > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python3 tools/overlap_report.py gpurun_out/prof_$TAG | tee gpurun_out/prof_${TAG}_overlap.txt
python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/prof_$TAG/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open("gpurun_out/prof_${TAG}_kernel_stats.txt", "w") as o:
    o.write("rocprofv3 --kernel-trace --stats of: python3 bench.py --no-cpu --no-sides --no-config3  (300 steps, 3 forwards in flight, HIP-graph replay)\n")
    for r in rows[:30]:
        o.write(f"{r['Name'][:110]:110s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %\n")
PY
rm -rf gpurun_out/prof_$TAG
