#!/bin/bash
# round 4, visit A: the new parity cases at the benchmarked shape + a baseline bench line of the round-3 kernels on today's box
tag=${1:-r4a}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1500 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 1200 -p no:cacheprovider -s > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/${tag}_${name}.log | cut -c1-400; }
run long_attn tests/test_gpu_kernels.py -k long_sequences
run opwise_model tests/test_gpu_opwise_model.py
run opwise_b8 "tests/test_gpu_opwise.py" -k "bf16-8"
run fullsize_b8 "tests/test_gpu_fullsize.py" -k "bf16-8"
timeout 900 python bench.py > gpurun_out/${tag}_bench_default.log 2>&1
echo "bench default rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_default.log | cut -c1-1500
echo "total t=$((SECONDS-t0))"
