#!/bin/bash
# round 6: key-split 8-wave dQ kernel (ALM_ATTN_DQ8=1) -- attention tests under it, timing A/B on one box
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
ALM_ATTN_DQ8=1 timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "mqa_attention and not either_dkv" > gpurun_out/r6q8_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/r6q8_tests.log
for r in 1 2; do
  for v in 0 1; do
    echo "== ALM_ATTN_DQ8=$v"
    ALM_ATTN_DQ8=$v timeout 600 python scripts/attn_bench.py 1024 2048 2049 8253 16385
  done
done > gpurun_out/r6q8_dq8_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r6q8_dq8_ab.log
