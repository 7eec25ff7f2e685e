#!/bin/bash
# round 4, visit C: hyper-connection backward diet (LDS scalar record, folded weights, v_dot2): correctness, then A/B against the previous build
tag=${1:-r4c}
prev=scripts/ubench/bin/libaudiolm_hip_prev.so
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1200 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_${name}.log | cut -c1-500; }
run hc_kernels tests/test_gpu_kernels.py -k "hyper_connections"
run opwise tests/test_gpu_opwise.py -k "None"
run parity_small tests/test_gpu_parity.py -x
for i in 1 2 3; do
  ALM_LIB_PATH=$prev python scripts/hc_bench.py 2>&1 | tail -1
  python scripts/hc_bench.py 2>&1 | tail -1
done | tee gpurun_out/${tag}_hc_ab.log
echo "hc A/B t=$((SECONDS-t0))"
bash scripts/ab_step.sh $prev 2 2>&1 | tee gpurun_out/${tag}_ab_step.log
echo "total t=$((SECONDS-t0))"
