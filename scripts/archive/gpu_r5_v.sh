#!/bin/bash
tag=${1:-r5v}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
for r in 1 2; do
  for e in "ALM_X=0" "ALM_GEMM_RING=0" "ALM_HEAD_KCAT=0" "ALM_PACK_ALL=0 ALM_PACK_WIDE=0" "ALM_EMBED_SCATTER=atomic" "ALM_GEMM_GROUP2=0" "ALM_QKV_GROUP=1"; do
    ms=$(env $e timeout 300 python bench.py --config coarse1024 --steps 30 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('roofline') or {}).get('all_gemm_launches', {}).get('by_kind_ms'))")
    echo "round $r [$e] $ms"
  done
done > gpurun_out/${tag}_coarse1024_ab.log 2>&1
cat gpurun_out/${tag}_coarse1024_ab.log
