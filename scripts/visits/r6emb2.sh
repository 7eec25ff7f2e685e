#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
B=$PWD/scripts/ubench/bin
for r in 1 2; do
  ALM_LIB_PATH=$B/libaudiolm_hip_prev.so python scripts/embed_scatter_bench.py 2>&1 | grep -v amdgpu
  python scripts/embed_scatter_bench.py 2>&1 | grep -v amdgpu
done | tee gpurun_out/r6emb2_scatter_bench.log
