#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_opwise_model.py tests/test_gpu_defaults.py tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "embed or scatter or deterministic or coarse" > gpurun_out/r6emb3_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/r6emb3_tests.log
