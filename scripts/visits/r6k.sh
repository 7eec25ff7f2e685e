#!/bin/bash
# round 6, visit k: strided down-sampling conv kernel -- codec tests + tokenize A/B
tag=${1:-r6k}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 900 python -m pytest tests/test_gpu_codec.py -m gpu -q --tb=short > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -n 12 gpurun_out/${tag}_tests.log
for e in "ALM_CONV_STRIDED=1" "ALM_CONV_STRIDED=0" "ALM_CONV_STRIDED=1" "ALM_CONV_STRIDED=0 ALM_FUSE_RESUNIT=0"; do echo "== $e"; env $e timeout 600 python scripts/conv_bench.py 2>&1 | tail -n 1; done > gpurun_out/${tag}_tokenize_ab.log 2>&1
cat gpurun_out/${tag}_tokenize_ab.log
