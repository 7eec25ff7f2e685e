#!/bin/bash
# launch-list visit: the bitwise tests, host issue split and step time with the lists on / off (interleaved)
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
T=${1:-r6g}
timeout 900 python -m pytest tests/test_gpu_launchlist.py -m gpu -q --tb=short -x --timeout 600 > gpurun_out/${T}_ll_tests.log 2>&1
echo "ll tests rc=$?"; tail -n 30 gpurun_out/${T}_ll_tests.log
for cfg in coarse2048 coarse1024; do
  for ll in 1 0 1 0; do
    echo "== host_split $cfg ALM_LAUNCH_LIST=$ll"
    ALM_LAUNCH_LIST=$ll timeout 300 python scripts/host_split.py $cfg 2>&1 | tail -n 2
  done
done > gpurun_out/${T}_host_split.log 2>&1
cat gpurun_out/${T}_host_split.log
for r in 1 2; do
  for ll in 1 0; do
    for cfg in coarse2048 coarse1024; do
      echo "== bench $cfg ALM_LAUNCH_LIST=$ll round $r"
      ALM_LAUNCH_LIST=$ll timeout 600 python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-optimizer-leg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        o=json.loads(l); print(o['ms_per_step'], o['host'], o['roofline']['frac'] if 'roofline' in o else None, o['loss'])
"
    done
  done
done > gpurun_out/${T}_bench_ab.log 2>&1
cat gpurun_out/${T}_bench_ab.log
