#!/usr/bin/env bash
# Same-box A/B on the headline: units per wave of the row-loop kernels (fewer, longer-lived workgroups = less chip time per launch)
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
run() {
  local label=$1; shift
  local v=$(env "$@" python bench.py --no-config3 --no-cpu --no-sides 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s, latency %.3f ms, swin stages %.3f ms' % (d['value'], d['latency_ms_single_stream'], d['attention_gemm']['swin_stages_ms']))")
  echo "$label: $v"
}
for rep in 1 2; do
  run "stages 1-3, r x 1" SDF_RES_RMUL=1
  run "stages 1-3, r x 2" SDF_RES_RMUL=2
  run "stages 1-3, r x 3" SDF_RES_RMUL=3
  run "stages 0-3, r x 2" SDF_RES_RMUL=2 SDF_RES_MINC=96
done
