#!/bin/bash
tag=${1:-r5j}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_opwise_model.py tests/test_gpu_parity.py tests/test_gpu_generate.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -x \
  -k "logit_heads or opwise or golden or deterministic or generate or greedy" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/${tag}_tests.log | cut -c1-300
STEPS=30 bash scripts/ab_env2.sh 3 "ALM_HEAD_KCAT=1" "ALM_HEAD_KCAT=0" > gpurun_out/${tag}_ab.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab.log | cut -c1-200
echo "total t=$((SECONDS-t0))"
