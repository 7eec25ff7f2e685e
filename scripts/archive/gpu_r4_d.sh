#!/bin/bash
# round 4, visit D: hyper-connection kernels with unconditional stores (no vmcnt(0) drains), readlane vs LDS-record element loop; optimizer table in kernargs
tag=${1:-r4d}
prev=scripts/ubench/bin/libaudiolm_hip_prev.so
ldsrec=scripts/ubench/bin/libaudiolm_hip_ldsrec.so
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1200 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_${name}.log | cut -c1-500; }
run hc_opt tests/test_gpu_kernels.py tests/test_gpu_optimizer.py -k "hyper_connections or adam"
ALM_LIB_PATH=$ldsrec run hc_ldsrec tests/test_gpu_kernels.py -k "hyper_connections"
run opwise tests/test_gpu_opwise.py -k "None"
run parity_small tests/test_gpu_parity.py -x
for i in 1 2 3; do
  ALM_LIB_PATH=$prev python scripts/hc_bench.py 2>&1 | tail -1
  ALM_LIB_PATH=$ldsrec python scripts/hc_bench.py 2>&1 | tail -1
  python scripts/hc_bench.py 2>&1 | tail -1
done | tee gpurun_out/${tag}_hc_ab.log
echo "hc A/B t=$((SECONDS-t0))"
run() { python bench.py --steps 20 --warmup 5 --schedule eager --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']['all_gemm_launches']['by_kind_ms']; k = {x['kernel']: x['ms_per_step'] for x in d['roofline']['kernels']}
print('$1', d['ms_per_step'], 'ms/step  hc', k.get('hc_fwd'), k.get('hc_bwd'), ' opt', d.get('with_optimizer'))"; }
for i in 1 2; do
  run default
  ALM_LIB_PATH=$prev run prev
  ALM_LIB_PATH=$ldsrec run ldsrec
done 2>&1 | tee gpurun_out/${tag}_ab_step.log
echo "total t=$((SECONDS-t0))"
