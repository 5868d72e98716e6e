#!/usr/bin/env bash
# Same-box A/B of the routing of the weight-resident row-loop kernels (csrc/ms_res.hip) on the HEADLINE (three forwards in flight):
# which stages they serve is a chip-time question, not a latency one.  usage (GPU box): tools/res_routing_ab.sh > gpurun_out/x.txt
cd "${GRAFT_REPO_ROOT:?run on the GPU box}" || exit 1
run() {  # label, env assignments...
  local label=$1; shift
  local v=$(env "$@" python bench.py --no-config3 --no-cpu --no-sides 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f samples/s, latency %.3f ms, swin stages %.3f ms' % (d['value'], d['latency_ms_single_stream'], d['attention_gemm']['swin_stages_ms']))")
  echo "$label: $v"
}
for rep in 1 2; do
  run "default (C in 128..192)        " SDF_RES_MINC=128 SDF_RES_MAXC=192
  run "stages 1-2 (C in 128..384)     " SDF_RES_MINC=128 SDF_RES_MAXC=384
  run "stages 1-3 (C in 128..768)     " SDF_RES_MINC=128 SDF_RES_MAXC=768
  run "stages 0-3 (C in 96..768)      " SDF_RES_MINC=96 SDF_RES_MAXC=768
  run "off (SDF_RES=0: round-4 kernels)" SDF_RES=0
done
