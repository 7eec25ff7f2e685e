#!/bin/bash
# round 6 visit q: NT big tiles, the bounded attempt of VERDICT item 6 -- one operand pinned in the XCD's L2 (tile-column groups), the other streamed with the
# non-temporal DMA policy; per configuration the step time and the GEMM time by kind (HIP events inside bench.py)
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
B=$PWD/scripts/ubench/bin
run() { env $2 python bench.py --steps 20 --warmup 5 --residual bf16 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']['all_gemm_launches']['by_kind_ms']
print('$1', d['ms_per_step'], 'ms/step  frac', d['roofline']['frac'], ' gemm by kind', r)"; }
for i in 1 2; do
  run default "X=1"
  run auxA2 "ALM_LIB_PATH=$B/libaudiolm_hip_gemm_auxA2.so"
  run auxB2 "ALM_LIB_PATH=$B/libaudiolm_hip_gemm_auxB2.so"
  run auxA2_colgroups3 "ALM_LIB_PATH=$B/libaudiolm_hip_gemm_auxA2.so ALM_GEMM_GROUP_M=-3"
  run colgroups3 "ALM_GEMM_GROUP_M=-3"
  run auxB2_rowgroups4 "ALM_LIB_PATH=$B/libaudiolm_hip_gemm_auxB2.so ALM_GEMM_GROUP_M=4"
done > gpurun_out/r6q_nt_policy_ab.log 2>&1
cat gpurun_out/r6q_nt_policy_ab.log
