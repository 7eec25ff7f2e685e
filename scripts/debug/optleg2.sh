pr() { tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['with_optimizer']['ms_per_step'], d['with_optimizer']['torch_adam_ms_per_step'], d['host'])"; }
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | pr "supervised,no-cpu"
ALM_BENCH_SUPERVISE=0 timeout 600 python bench.py 2>/dev/null | pr "inplace,cpu-baseline"
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | pr "supervised,no-cpu"
