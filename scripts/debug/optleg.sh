export ALM_BENCH_SUPERVISE=0
for cfg in "ALM_DEFER_WGRAD=1" "ALM_DEFER_WGRAD=0" "ALM_DEFER_WGRAD=1"; do
  env $cfg timeout 300 python bench.py --steps 20 --warmup 5 --schedule eager --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['with_optimizer']['ms_per_step'], d['with_optimizer']['torch_adam_ms_per_step'], d['host'])"
done
