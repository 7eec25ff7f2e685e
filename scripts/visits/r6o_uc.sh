#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_opwise_model.py -m gpu -q -x --tb=short -p no:cacheprovider -k "unique or prepare or wrapper or golden or bookkeeping or fused or coarse or semantic" > gpurun_out/r6o_tests.log 2>&1
echo "tests rc=$?"; tail -n 12 gpurun_out/r6o_tests.log | cut -c1-300
timeout 500 python scripts/uc_bench.py 30 2>&1 | grep "round" | tee gpurun_out/r6o_uc_after.log
