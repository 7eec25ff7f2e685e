#!/bin/bash
# round 6, visit j: time blocks per wave of the fused ResidualUnit (ALM_RESUNIT_NJ A/B)
tag=${1:-r6j}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
for nj in 0 1 2 4; do echo "== ALM_RESUNIT_NJ=$nj"; ALM_RESUNIT_NJ=$nj timeout 600 python scripts/conv_bench.py 2>&1 | grep -v amdgpu; done > gpurun_out/${tag}_conv_nj.log 2>&1
cat gpurun_out/${tag}_conv_nj.log
