export PYTHONUNBUFFERED=1 TMPDIR=/tmp
pr() { tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['with_optimizer']['ms_per_step'], d['with_optimizer']['torch_adam_ms_per_step'], d['host'])"; }
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | pr "fresh box"
timeout 600 python -m pytest tests/test_gpu_dp.py tests/test_gpu_graphed.py tests/test_gpu_generate.py -m gpu -q 2>&1 | tail -1
ps aux | grep -c "[p]ython"; ps aux --sort=-%cpu | head -5 | cut -c1-150
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | pr "after dp/graphed/generate tests"
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -1
free -g | head -2
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | pr "after fullsize tests"
