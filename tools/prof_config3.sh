#!/usr/bin/env bash
# rocprofv3 --kernel-trace --stats of BASELINE config 3 (tools/ann_try.py: STTFlowNet, batch 8, 288 x 384) -> gpurun_out/prof_$1
set -e
TAG=${1:-c3}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/tools/ann_try.py > $R/gpurun_out/prof_$TAG.log 2>&1 || true
cd $R
grep STTFlowNet gpurun_out/prof_$TAG.log | tail -1
python3 - <<PY
import csv, glob
fs = glob.glob("gpurun_out/prof_$TAG/**/*_kernel_stats.csv", recursive=True)
if not fs:
    raise SystemExit("no kernel_stats.csv")
rows = list(csv.DictReader(open(fs[0])))
print("rocprofv3 --kernel-trace --stats of: python3 tools/ann_try.py  (15 forwards of config 3: 2 checked + 3 warm-up + 10 timed)")
for r in rows:                                        # (every kernel of the run: library remnants included)
    print(f"{r['Name'][:120]:120s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:8.1f} us  {float(r['Percentage']):5.1f} %")
PY
