pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%8.1f samples/s  %.3f ms/step  latency %.3f ms" % (d["value"], d["ms_per_step"], d["latency_ms_single_stream"]))'
for f in 3 4 5 6 3; do echo -n "inflight $f: "; python3 bench.py --no-cpu --no-sides --no-config3 --inflight $f 2>/dev/null | python3 -c "$pick"; done
