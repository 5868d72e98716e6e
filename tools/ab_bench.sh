#!/usr/bin/env bash
# same-box A/B of the default bench line: this tree against a baseline checkout under .ab_base/ (a git worktree of an older
# commit, built in place; git-ignored), run alternately.  usage (GPU box): tools/ab_bench.sh [rounds] [extra bench.py args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-2}
shift || true
cd $R
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%8.1f samples/s  %.3f ms/step  latency %.3f ms  swin stages %.3f ms" % (d["value"], d["ms_per_step"], d["latency_ms_single_stream"], d["attention_gemm"]["swin_stages_ms"]))'
for i in $(seq $N); do
  echo -n "base : "; (cd .ab_base && python3 bench.py --no-cpu "$@" 2>/dev/null | python3 -c "$pick")
  echo -n "this : "; python3 bench.py --no-cpu --no-sides "$@" 2>/dev/null | python3 -c "$pick"
done
