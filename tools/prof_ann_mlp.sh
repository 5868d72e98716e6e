#!/usr/bin/env bash
# rocprofv3 kernel statistics of the MLP half block of config 3's first stage, one launch vs three (tools/ann_block_one.py ... mlp)
TAG=${1:-r5}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets GRAFT_REPO_ROOT)}" || exit 1
OUT=gpurun_out/prof_ann_mlp
: > gpurun_out/${TAG}_ann_mlp_stats.txt
for w in fused four; do
  rm -rf "${OUT:?}"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o p -- python3 tools/ann_block_one.py plain $w mlp 2>/dev/null | grep "half block" >> gpurun_out/${TAG}_ann_mlp_stats.txt
  f=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> gpurun_out/${TAG}_ann_mlp_stats.txt <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:5]:
    print(f"    {float(r['AverageNs'])/1e3:8.1f} us avg x {int(r['Calls']):4d}  {r['Name'][:110]}")
PY
done
cat gpurun_out/${TAG}_ann_mlp_stats.txt
