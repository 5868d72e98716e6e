#!/usr/bin/env bash
# rocprofv3 kernel trace of eager single-stream forwards of config 2 (tools/forward_one.py) -> gpurun_out/prof_$1 + the launch
# sequence of one forward with a per-kernel summary (tools/prof_seq2.py)
set -e
TAG=${1:-fwd}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python3 $R/tools/forward_one.py 8 > $R/gpurun_out/prof_$TAG.log 2>&1
cd $R
python3 tools/prof_seq2.py gpurun_out/prof_$TAG > gpurun_out/prof_${TAG}_sequence.txt
tail -45 gpurun_out/prof_${TAG}_sequence.txt
rm -rf gpurun_out/prof_$TAG
