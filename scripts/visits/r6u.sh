#!/bin/bash
# round 6 visit u: to_q || to_kv grouped launch per shape class (ALM_QKV_GROUP 0 / 1 / auto) on configs[1] (M = 8192) and the headline (M = 16384), interleaved
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
run() { env $3 timeout 300 python bench.py --config $1 --steps 40 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$1 [$2]', d['ms_per_step'], d['roofline']['all_gemm_launches']['by_kind_ms'])"; }
for r in 1 2 3; do
  for cf in coarse1024 coarse2048; do
    run $cf "ALM_QKV_GROUP=0" "ALM_QKV_GROUP=0"
    run $cf "ALM_QKV_GROUP=1" "ALM_QKV_GROUP=1"
    run $cf "auto" "X=1"
  done
done > gpurun_out/r6u_qkv_group_ab.log 2>&1
cat gpurun_out/r6u_qkv_group_ab.log
