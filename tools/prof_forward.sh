#!/usr/bin/env bash
# rocprofv3 kernel trace of bench.py on the GPU box -> gpurun_out/prof_$1 (+ per-forward summary and launch sequence)
set -e
TAG=${1:-r1}
shift || true                      # further arguments go to bench.py (e.g. --inflight 1)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/bench.py --no-cpu --steps 10 "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python3 tools/prof_summary.py gpurun_out/prof_$TAG > gpurun_out/prof_${TAG}_summary.txt
python3 tools/prof_sequence.py gpurun_out/prof_$TAG > gpurun_out/prof_${TAG}_sequence.txt
cat gpurun_out/prof_${TAG}_summary.txt
