#!/bin/bash
# round 6 visit t: dK/dV kernel with 2 heads per workgroup (two workgroups per CU, independent barriers) vs 4: kernel tests under both, timing A/B on one box
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
for nh in 2 4; do
  ALM_ATTN_DKV_NH=$nh timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bias.py tests/test_gpu_dropout.py -m gpu -q --tb=short -x -k "mqa or attention or bias or dropout" > gpurun_out/r6t_tests_nh$nh.log 2>&1
  echo "tests NH=$nh rc=$?"; tail -n 3 gpurun_out/r6t_tests_nh$nh.log
done
for r in 1 2; do
  for nh in 4 2; do
    echo "== ALM_ATTN_DKV_NH=$nh"
    ALM_ATTN_DKV_NH=$nh timeout 600 python scripts/attn_bench.py 1024 2048 2049 8253 16385
  done
done > gpurun_out/r6t_dkv_nh_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r6t_dkv_nh_ab.log
