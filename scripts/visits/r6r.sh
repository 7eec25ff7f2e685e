#!/bin/bash
# round 6 visit r: new kernel tests (ring-tile epilogues, owned-scatter fallback), optimizer tests (prepared-table key), forward QB = 2 for long sequences A/B
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_optimizer.py -m gpu -q --tb=short -x -n 4 > gpurun_out/r6r_tests.log 2>&1
echo "tests rc=$?"; tail -n 5 gpurun_out/r6r_tests.log
for r in 1 2; do
  timeout 600 python scripts/attn_bench.py 2048 8253 16385
  ALM_ATTN_FWD_QB2_MINN=2048 timeout 600 python scripts/attn_bench.py 2048 8253 16385
done > gpurun_out/r6r_fwd_qb2_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r6r_fwd_qb2_ab.log
