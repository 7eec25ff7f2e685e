#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python scripts/debug/memset_node_probe.py 2>&1 | grep bytes | tee gpurun_out/r6v_memset_node_probe.log
timeout 1200 python -m pytest tests/test_gpu_graphed.py tests/test_gpu_launchlist.py tests/test_gpu_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "graphed or launch or hc or hyper or embed or scatter or memset" 2>&1 | tail -4 | cut -c1-200
