#!/bin/bash
# round 6, visit d: attention A/B (round-5 attention.hip vs this tree) at the timed shapes on one box + the dK/dV segment probe + kernel tests
tag=${1:-r6d}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "mqa or attention" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -n 8 gpurun_out/${tag}_tests.log
for r in 1 2; do
  ALM_LIB_PATH=$PWD/scripts/ubench/bin/libaudiolm_hip_attn_r5.so timeout 600 python scripts/attn_bench.py 1024 2048 2049 8253 16385
  timeout 600 python scripts/attn_bench.py 1024 2048 2049 8253 16385
done > gpurun_out/${tag}_attn_ab.log 2>&1
cat gpurun_out/${tag}_attn_ab.log
ALM_PROBE_KERNEL=dkv timeout 300 python scripts/attn_probe.py run > gpurun_out/${tag}_dkv_probe.log 2>&1
cat gpurun_out/${tag}_dkv_probe.log
