#!/bin/bash
# round 6, visit i: codec after the occupancy hints -- tests + tokenize timing + e2e_config5 bench line with per-kernel numbers
tag=${1:-r6i}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 900 python -m pytest tests/test_gpu_codec.py -m gpu -q --tb=short -x > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/${tag}_tests.log
timeout 600 python scripts/conv_bench.py 2>&1 | tail -n 1 > gpurun_out/${tag}_conv_bench.log
cat gpurun_out/${tag}_conv_bench.log
timeout 900 python bench.py --config e2e_config5 --steps 5 --warmup 2 --no-optimizer-leg > gpurun_out/${tag}_bench_e2e.log 2>&1
tail -n 1 gpurun_out/${tag}_bench_e2e.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], d.get('parity'))
for k in d['roofline']['kernels']: print(k['kernel'], k['launches_per_step'], k['ms_per_step'], k['frac'])
"
