#!/bin/bash
# same-box check that the round-5 defaults do not regress the other configurations: defaults vs every round-5 switch reverted
tag=${1:-r5u}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
OLD="ALM_GEMM_RING=0 ALM_PACK_ALL=0 ALM_PACK_WIDE=0 ALM_HEAD_KCAT=0 ALM_EMBED_SCATTER=atomic ALM_GEMM_GROUP2=0"
for cf in coarse1024 fine2049; do
  for r in 1 2 3; do
    for v in new old; do
      if [[ $v == old ]]; then e="$OLD"; else e="ALM_X=0"; fi
      ms=$(env $e timeout 300 python bench.py --config $cf --steps 30 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host'])")
      echo "$cf round $r [$v] $ms"
    done
  done
done > gpurun_out/${tag}_configs_ab.log 2>&1
cat gpurun_out/${tag}_configs_ab.log
