#!/usr/bin/env bash
# alternating same-box runs of the headline at several numbers of forwards in flight (bench.py --inflight): usage inflight_ab2.sh "3 6 3 6 9 3"
pick='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%8.1f samples/s  %.3f ms/step  latency %.3f ms" % (d["value"], d["ms_per_step"], d["latency_ms_single_stream"]))'
for f in ${1:-3 6 3 6 9 3 6}; do echo -n "inflight $f: "; python3 bench.py --no-cpu --no-sides --no-config3 --inflight $f 2>/dev/null | python3 -c "$pick"; done
