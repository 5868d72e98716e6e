# same-box A/B of environment switches on the default bench line.  usage: tools/cfg_ab.sh VAR v1 v2 ...   ("none" = unset)
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%8.1f samples/s  %.3f ms/step  latency %.3f ms  swin stages %.3f ms" % (d["value"], d["ms_per_step"], d["latency_ms_single_stream"], d["attention_gemm"]["swin_stages_ms"]))'
VAR=$1; shift
for c in "$@"; do
  echo -n "$VAR=$c : "
  if [ $c = none ]; then python3 bench.py --no-cpu --no-sides --no-config3 2>/dev/null | python3 -c "$pick"; else env $VAR=$c python3 bench.py --no-cpu --no-sides --no-config3 2>/dev/null | python3 -c "$pick"; fi
done
