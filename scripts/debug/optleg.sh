# the `with_optimizer` leg of bench.py under different flows (debug: an outlier of 14.3 ms was seen in one default run)
export ALM_BENCH_SUPERVISE=0
for cfg in "--schedule eager" "--schedule auto" "--schedule eager" "--schedule auto"; do
  timeout 300 python bench.py --steps 20 --warmup 5 $cfg --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['with_optimizer']['ms_per_step'], d['with_optimizer']['torch_adam_ms_per_step'], d['host'], d['config'].get('schedule_probe'))"
done
