#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_launchlist.py -m gpu -q -x --tb=short -p no:cacheprovider -k "graphed or unique or prepare or fused or golden or launch" 2>&1 | tail -3 | cut -c1-200
timeout 600 python scripts/uc_bench.py 30 2>&1 | grep round | tee gpurun_out/r6x_uc_prepack.log
