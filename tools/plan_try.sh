cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for plan in "" "0:10,1:5,1:5" "0:10,1:4,1:6" "0:5,1:3,0:5,1:7" "0:7,1:3,1:7,0:3" "0:10,1:3,1:7"; do
  python bench.py --steps 20 --warmup 5 --no-sides --no-config3 --no-cpu ${plan:+--plan $plan} 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().split('\n')[-1]); h=r['headline_summary']
print('plan [$plan]: value(20 steps) %.1f  over96 %.1f' % (h['value'], h['value_over_90_steps']['samples_per_s_this_rank']))"
done; done
