#!/bin/bash
tag=${1:-r5f}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider -k "gemm_nt_tile_configs" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 3 gpurun_out/${tag}_tests.log | cut -c1-300
STEPS=30 bash scripts/ab_env2.sh 3 "ALM_GEMM_MID_TILE=0" "ALM_GEMM_MID_TILE=1" > gpurun_out/${tag}_ab.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab.log | cut -c1-250
echo "total t=$((SECONDS-t0))"
