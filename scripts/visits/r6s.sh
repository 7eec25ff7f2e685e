#!/bin/bash
# round 6 visit s: s_setprio around the MFMA clusters of the flash forward / dQ kernels (measurement build) vs the shipped library, one box, interleaved
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
B=$PWD/scripts/ubench/bin
for r in 1 2; do
  ALM_LIB_PATH=$B/libaudiolm_hip_r6x.so timeout 600 python scripts/attn_bench.py 1024 2048 8253 16385
  ALM_LIB_PATH=$B/libaudiolm_hip_attn_prio.so timeout 600 python scripts/attn_bench.py 1024 2048 8253 16385
done > gpurun_out/r6s_attn_prio_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r6s_attn_prio_ab.log | sed 's#/root/repo/scripts/ubench/bin/##'
