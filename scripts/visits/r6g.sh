#!/bin/bash
# round 6, visit g: fused ResidualUnit -- tests, per-stage micro-benchmark, tokenize A/B
tag=${1:-r6g}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 900 python -m pytest tests/test_gpu_codec.py -m gpu -q --tb=short -x > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/${tag}_tests.log
timeout 600 python scripts/conv_bench.py > gpurun_out/${tag}_conv_bench.log 2>&1
ALM_RESUNIT_NJ64=1 timeout 600 python scripts/conv_bench.py 2>&1 | grep "C=  64\|tokenize" >> gpurun_out/${tag}_conv_bench.log
ALM_FUSE_RESUNIT=0 timeout 600 python scripts/conv_bench.py 2>&1 | tail -n 1 >> gpurun_out/${tag}_conv_bench.log
cat gpurun_out/${tag}_conv_bench.log
