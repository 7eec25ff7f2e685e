#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0 SCATTER_CASES=headline
mkdir -p gpurun_out
prev=$PWD/audiolm-pytorch_amd/libaudiolm_hip_prev.so
for v in prev new; do
  if [[ $v == prev ]]; then export ALM_LIB_PATH=$prev; else unset ALM_LIB_PATH; fi
  rm -rf /tmp/prof_$v; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o k -- python scripts/embed_scatter_bench.py > /dev/null 2>&1
  db=$(find /tmp/prof_$v -name "*.db" | head -1); python scripts/prof_summary.py "$db" gpurun_out/r6l_scatter_kernels_$v.csv "embed_scatter_bench headline uniform ($v)" | tail -1
  head -6 gpurun_out/r6l_scatter_kernels_$v.csv | cut -c1-120,240-330
done
unset ALM_LIB_PATH SCATTER_CASES
log=gpurun_out/r6l_step_ab.log; : > $log
for r in 1 2 3 4; do
  for lib in new prev prev new; do
    if [[ $lib == prev ]]; then export ALM_LIB_PATH=$prev; else unset ALM_LIB_PATH; fi
    ms=$(timeout 600 python bench.py --steps 40 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('loss'))")
    echo "round $r coarse2048 [$lib] $ms" | tee -a $log
  done
done
