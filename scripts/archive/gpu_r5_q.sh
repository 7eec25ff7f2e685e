#!/bin/bash
tag=${1:-r5q}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
for r in 1 2 3; do
  for sch in eager graph; do
    ms=$(timeout 300 python bench.py --steps 40 --warmup 8 --schedule $sch --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['schedule'][:20], d['host'])")
    echo "round $r [$sch] $ms"
  done
done > gpurun_out/${tag}_sched.log 2>&1
cat gpurun_out/${tag}_sched.log
