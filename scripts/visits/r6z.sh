#!/bin/bash
# round 6 visit z: SPLIT-Q of the dK/dV kernel for launches of <= one workgroup per CU: tests (auto and forced off), timing at short sequences, configs[1] step
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
for sp in 0 1; do
  ALM_ATTN_DKV_SPLIT=$sp timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bias.py tests/test_gpu_dropout.py -m gpu -q --tb=short -x -k "(mqa or attention or bias or dropout) and not either_dkv" > gpurun_out/r6z_tests_sp$sp.log 2>&1
  echo "tests ALM_ATTN_DKV_SPLIT=$sp rc=$?"; tail -n 3 gpurun_out/r6z_tests_sp$sp.log
done
for r in 1 2; do
  for sp in 1 2; do
    echo "== ALM_ATTN_DKV_SPLIT=$sp"
    ALM_ATTN_DKV_SPLIT=$sp timeout 600 python scripts/attn_bench.py 512 1024 1536 2048
  done
done > gpurun_out/r6z_dkv_split_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r6z_dkv_split_ab.log
for r in 1 2 3; do
  for sp in 1 0; do
    ALM_ATTN_DKV_SPLIT=$sp timeout 300 python bench.py --config coarse1024 --steps 40 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('coarse1024 [ALM_ATTN_DKV_SPLIT=$sp]', d['ms_per_step'], {k['kernel']: k['ms_per_step'] for k in d['roofline']['kernels'] if 'mqa' in k['kernel']})"
  done
done 2>&1 | tee -a gpurun_out/r6z_dkv_split_ab.log
