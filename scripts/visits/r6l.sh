#!/bin/bash
# round 6, visit l: RVQ with 8 waves per workgroup -- codec tests + tokenize A/B
tag=${1:-r6l}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
for w in 8 4; do ALM_RVQ_WAVES=$w timeout 900 python -m pytest tests/test_gpu_codec.py -m gpu -q --tb=short -k "rvq or golden or soundstream" 2>&1 | tail -n 2; done > gpurun_out/${tag}_tests.log 2>&1
cat gpurun_out/${tag}_tests.log
for e in "ALM_RVQ_WAVES=4" "ALM_RVQ_WAVES=8" "ALM_RVQ_WAVES=4" "ALM_RVQ_WAVES=8"; do echo "== $e"; env $e timeout 600 python scripts/conv_bench.py 2>&1 | tail -n 1; done > gpurun_out/${tag}_tokenize_ab.log 2>&1
cat gpurun_out/${tag}_tokenize_ab.log
