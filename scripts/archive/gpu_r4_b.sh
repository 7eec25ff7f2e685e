#!/bin/bash
# round 4, visit B: re-check of the fixed / new tests (dQ block map, model-level op-wise, graph dropout counter, autocast default, grouped DP path)
tag=${1:-r4b}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1500 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 1200 -p no:cacheprovider -s > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 5 gpurun_out/${tag}_${name}.log | cut -c1-600; }
run attn tests/test_gpu_kernels.py -k "attention"
run opwise_model tests/test_gpu_opwise_model.py
run graphed_defaults_dropout tests/test_gpu_graphed.py tests/test_gpu_defaults.py tests/test_gpu_dropout.py
run dp tests/test_gpu_dp.py
echo "total t=$((SECONDS-t0))"
