#!/bin/bash
# round 4, visit J: hc_bwd on the LDS-DMA path (ALM_HC_GL=1) vs the register prefetch; FusedAdam + pack (visit I's content)
tag=${1:-r4j}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1200 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/${tag}_${name}.log | cut -c1-600; }
ALM_HC_GL=1 run hc_gl tests/test_gpu_kernels.py -k "hyper_connections"
ALM_HC_GL=0 run hc_reg tests/test_gpu_kernels.py -k "hyper_connections"
ALM_HC_GL=1 run opwise_gl tests/test_gpu_opwise.py -k "coarse-4-bf16-None or fine-4-bf16-None"
for i in 1 2 3; do
  ALM_HC_GL=0 python scripts/hc_bench.py 2>&1 | tail -1
  ALM_HC_GL=1 python scripts/hc_bench.py 2>&1 | tail -1
done | tee gpurun_out/${tag}_hc_ab.log
echo "hc A/B t=$((SECONDS-t0))"
run opt tests/test_gpu_optimizer.py tests/test_gpu_defaults.py
leg() { python bench.py --steps 20 --warmup 5 --schedule eager --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); o = d['with_optimizer']; k = {x['kernel']: x['ms_per_step'] for x in d['roofline']['kernels']}
print('$1', d['ms_per_step'], 'ms/step  hc', k.get('hc_fwd'), k.get('hc_bwd'), ' with_optimizer', o['ms_per_step'], 'torch', o['torch_adam_ms_per_step'], o['diagnostics']['fused']['optimizer_gpu_ms_median'])"; }
for i in 1 2; do
  ALM_HC_GL=0 ALM_FUSED_ADAM_PACK=1 leg reg_fusedpack
  ALM_HC_GL=1 ALM_FUSED_ADAM_PACK=1 leg gl__fusedpack
  ALM_HC_GL=0 ALM_FUSED_ADAM_PACK=0 leg reg_separate_
done 2>&1 | tee gpurun_out/${tag}_step_ab.log
echo "total t=$((SECONDS-t0))"
