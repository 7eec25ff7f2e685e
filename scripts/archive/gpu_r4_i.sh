#!/bin/bash
# round 4, visit I: FusedAdam writes the packed bf16 weight images (alm_opt_adam_pack_step): tests + the with_optimizer leg A/B
tag=${1:-r4i}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1200 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/${tag}_${name}.log | cut -c1-600; }
run opt tests/test_gpu_optimizer.py tests/test_gpu_defaults.py
run graphed_dp tests/test_gpu_graphed.py tests/test_gpu_dp.py -x
leg() { python bench.py --steps 20 --warmup 5 --schedule eager --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); o = d['with_optimizer']
print('$1', d['ms_per_step'], 'ms/step  with_optimizer', o['ms_per_step'], 'torch', o['torch_adam_ms_per_step'], o['diagnostics']['fused'])"; }
for i in 1 2; do
  ALM_FUSED_ADAM_PACK=1 leg fusedpack
  ALM_FUSED_ADAM_PACK=0 leg separate
done 2>&1 | tee gpurun_out/${tag}_opt_ab.log
echo "total t=$((SECONDS-t0))"
