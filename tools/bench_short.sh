#!/usr/bin/env bash
# The driver's command shape (python bench.py --steps 20 --warmup 5) for several (replicas, streams) settings: value, value over 96 steps, sustained.
# usage (GPU box): tools/bench_short.sh "R:F R:F ..." [repeats = 2]
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
for rep in $(seq 1 ${2:-2}); do
  for rf in $1; do
    R=${rf%%:*}; F=${rf##*:}
    python bench.py --steps 20 --warmup 5 --no-sides --no-config3 --no-cpu --replicas $R --inflight $F 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().split('\n')[-1]); h=r['headline_summary']
print('R=$R F=$F: value(20 steps) %.1f  over96 %.1f  sustained %.1f  latency %.2f ms' % (h['value'], h['value_over_90_steps']['samples_per_s_this_rank'], h['value_sustained_2s'], h['latency_ms_single_stream']))"
  done
done
