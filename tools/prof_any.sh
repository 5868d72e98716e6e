#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of any tools/*.py driver: per-kernel totals -> gpurun_out/prof_<tag>_kernel_stats.txt
# usage (GPU box): tools/prof_any.sh <tag> tools/cfg_try.py 4 20 480 640
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
PROG=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $PROG "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
tail -2 gpurun_out/prof_$TAG.log
python3 - <<PY | tee gpurun_out/prof_${TAG}_kernel_stats.txt
import csv, glob
f = glob.glob("gpurun_out/prof_$TAG/**/*_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"rocprofv3 --kernel-trace --stats of: python3 $PROG $@ ; kernel time {tot / 1e6:.1f} ms")
for r in rows[:32]:
    print(f"{r['Name'][:120]:120s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
rm -rf gpurun_out/prof_$TAG
