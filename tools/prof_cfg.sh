#!/usr/bin/env bash
# rocprofv3 kernel statistics of the SNN en4 forward at another BASELINE configuration (tools/cfg_try.py arguments)
# usage: tools/prof_cfg.sh TAG B T H W [wh ww] [lif|psn]   -> gpurun_out/TAG_cfg_stats.txt (top kernels by total time)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
OUT=gpurun_out/prof_cfg
rm -rf "${OUT:?}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- python3 tools/cfg_try.py "$@" 2>/dev/null | tail -1 > gpurun_out/${TAG}_cfg_stats.txt
f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> gpurun_out/${TAG}_cfg_stats.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time of the run {tot / 1e6:.1f} ms ({len(rows)} kernels); 9 forwards (2 + 2 warm-up + 5 timed)")
for r in rows[:28]:
    print(f"  {float(r['TotalDurationNs']) / 9e3:9.1f} us per forward  {float(r['AverageNs'])/1e3:8.1f} us avg x {int(r['Calls']) / 9:6.1f}  {r['Name'][:120]}")
PY
cat gpurun_out/${TAG}_cfg_stats.txt
