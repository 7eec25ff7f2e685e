#!/bin/bash
# round 6, visit b: more tile options on the under-filled NT shapes, the grouped dgrad pair on the big tile, attention micro-benchmarks at the timed shapes
tag=${1:-r6b}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 600 python scripts/ab_nt_inl.py 8192 16384 > gpurun_out/${tag}_nt_inl_cost.log 2>&1
cat gpurun_out/${tag}_nt_inl_cost.log
for e in "ALM_GEMM_GROUP2_BIG=0" "ALM_GEMM_GROUP2_BIG=1" "ALM_GEMM_GROUP2=0"; do env $e timeout 300 python scripts/ab_group2.py 8192 16384; done > gpurun_out/${tag}_group2.log 2>&1
cat gpurun_out/${tag}_group2.log
timeout 600 python scripts/attn_bench.py > gpurun_out/${tag}_attn_bench.log 2>&1
tail -n 40 gpurun_out/${tag}_attn_bench.log
