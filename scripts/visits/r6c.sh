#!/bin/bash
# round 6, visit c: the new parity cases at the timed batch shapes + the codebook-4096 op-wise row
tag=${1:-r6c}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_opwise_model.py -m gpu -q --tb=short -s -k "8-bf16 or 8253 or coarse4096 or 1024-bf16-8 or 2049-bf16-8" > gpurun_out/${tag}_fullsize.log 2>&1
echo "fullsize rc=$?"; grep -v "^  grad\|^   " gpurun_out/${tag}_fullsize.log | tail -n 40
