#!/bin/bash
# round 6 visit p: dQ kernel with packed score arithmetic + uniform diagonal branch: kernel tests, A/B against the previous build (r6m) on one box
tag=${1:-r6p}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "mqa or attention" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -n 8 gpurun_out/${tag}_tests.log
for r in 1 2; do
  ALM_LIB_PATH=$PWD/scripts/ubench/bin/libaudiolm_hip_r6m.so timeout 600 python scripts/attn_bench.py 1024 2048 2049 8253 16385
  timeout 600 python scripts/attn_bench.py 1024 2048 2049 8253 16385
done > gpurun_out/${tag}_attn_ab.log 2>&1
cat gpurun_out/${tag}_attn_ab.log
