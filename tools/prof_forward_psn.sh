#!/usr/bin/env bash
# as prof_forward_one.sh, with the shipped PSN neuron (configs/train_DSEC_supervised_SDformerFlow_en4.yml:49)
set -e
TAG=${1:-fwdpsn}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_$TAG"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/prof_$TAG" -o "$TAG" -- python3 "$R/tools/forward_one.py" 8 psn > "$R/gpurun_out/prof_$TAG.log" 2>&1
cd "$R"
python3 tools/prof_seq2.py "gpurun_out/prof_$TAG" > "gpurun_out/prof_${TAG}_sequence.txt"
tail -45 "gpurun_out/prof_${TAG}_sequence.txt"
rm -rf "gpurun_out/prof_$TAG"
