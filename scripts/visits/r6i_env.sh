#!/bin/bash
# A/B of HIP runtime switches (kernel-argument placement, hardware queue count) inside the timed step, interleaved on one box
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
log=gpurun_out/r6i_env_ab.log; : > $log
for r in 1 2 3; do
  for cf in coarse2048 coarse1024; do
    for cfg in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "GPU_MAX_HW_QUEUES=2"; do
      ms=$(env $cfg timeout 300 python bench.py --config $cf --steps 40 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host'].get('eager_issue_ms'))")
      echo "round $r  $cf [$cfg]  $ms" | tee -a $log
    done
  done
done
