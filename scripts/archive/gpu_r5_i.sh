#!/bin/bash
# round 5, visit I: token-major residual streams ([B][N][S][D], measurement build of hyper.hip) vs the shipped [B][S][N][D] -- stand-alone kernels, the
# step, and an end-to-end correctness check of the variant (the layout is internal to the stack: losses / gradients must not move)
tag=${1:-r5i}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
V=$PWD/scripts/ubench/bin/libaudiolm_hip_tokmaj.so
t0=$SECONDS
for r in 1 2 3; do
  python scripts/hc_bench.py 2>&1 | tail -n 1
  ALM_LIB_PATH=$V python scripts/hc_bench.py 2>&1 | tail -n 1
done > gpurun_out/${tag}_hc_ab.log 2>&1
cat gpurun_out/${tag}_hc_ab.log | sed 's|/tmp/[^ ]*/||'
ALM_LIB_PATH=$V timeout 900 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x \
  -k "full_size or golden or coarse" > gpurun_out/${tag}_variant_tests.log 2>&1
echo "variant tests rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_variant_tests.log | cut -c1-300
STEPS=30 bash scripts/ab_env2.sh 3 "ALM_X=0" "ALM_LIB_PATH=$V" > gpurun_out/${tag}_ab_step.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab_step.log | cut -c1-120 | sed 's|/tmp/[^ ]*/||'
echo "total t=$((SECONDS-t0))"
