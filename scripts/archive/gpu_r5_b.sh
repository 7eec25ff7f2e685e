#!/bin/bash
# round 5, visit B: owned scatter v2 (pipelined scan) + tests; GEMM rasterisation group / non-temporal C stores A/B in the step
tag=${1:-r5b}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider \
  -k "embed_scatter or embed_assemble or bitwise_deterministic or default_ctor_full_size or gemm_nt_tile_configs" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 8 gpurun_out/${tag}_tests.log | cut -c1-300
for g in 4 16 -4 -8; do
  ALM_GEMM_GROUP_M=$g timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm_nt" 2>&1 | tail -n 1
done
STEPS=30 bash scripts/ab_env2.sh 2 "ALM_X=0" "ALM_EMBED_SCATTER=atomic" "ALM_GEMM_GROUP_M=4" "ALM_GEMM_GROUP_M=16" "ALM_GEMM_GROUP_M=-4" "ALM_GEMM_GROUP_M=-8" "ALM_GEMM_NT_STORE=1" "ALM_GEMM_NT_STORE=2" "ALM_GEMM_GROUP_M=4 ALM_GEMM_NT_STORE=1" > gpurun_out/${tag}_ab.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab.log | cut -c1-400
echo "total t=$((SECONDS-t0))"
