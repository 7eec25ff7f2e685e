#!/bin/bash
# round 3, visit C: the whole GPU suite (attention dropout, standalone modules, RCCL 1-rank, re-scoped rounding-matched bounds) + smoke
tag=$1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 600 python -X faulthandler -m pytest tests/test_gpu_dropout.py tests/test_gpu_dp.py -m gpu -q --tb=short --timeout 500 > gpurun_out/${tag}_dropout_dp.log 2>&1
echo "dropout+dp rc=$? t=$((SECONDS-t0))"; tail -n 40 gpurun_out/${tag}_dropout_dp.log | cut -c1-400
rm -f gpurun_out/r3_fullsize_parity.jsonl gpurun_out/r3_opwise_parity.jsonl
timeout 2400 python -X faulthandler -m pytest tests -m gpu -q --tb=short --timeout 900 --deselect tests/test_gpu_dropout.py --deselect tests/test_gpu_dp.py -p no:cacheprovider > gpurun_out/${tag}_gpu_tests.log 2>&1
echo "all gpu tests rc=$? t=$((SECONDS-t0))"; tail -n 30 gpurun_out/${tag}_gpu_tests.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > gpurun_out/${tag}_smoke.log 2>&1
echo "smoke rc=$? t=$((SECONDS-t0))"; tail -n 3 gpurun_out/${tag}_smoke.log
echo "total t=$((SECONDS-t0))"
