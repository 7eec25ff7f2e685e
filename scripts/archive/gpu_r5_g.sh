#!/bin/bash
tag=${1:-r5g}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_dp.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > gpurun_out/${tag}_tests.log 2>&1
echo "dp tests rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/${tag}_tests.log | cut -c1-300
# python bench.py --gpus 2 with NO launcher: starts its own 2 ranks (both on cuda:0 over gloo: control flow only, the numbers mean nothing)
ALM_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_dp2_selflaunch.log 2>&1
echo "self-launch dp2 rc=$? t=$((SECONDS-t0))"; grep -E "launching|dp:" gpurun_out/${tag}_dp2_selflaunch.log | cut -c1-400; tail -n 1 gpurun_out/${tag}_dp2_selflaunch.log | cut -c1-1500
ALM_BENCH_SHARE_GPU=1 ALM_DP_DIRECT=0 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-optimizer-leg 2>&1 | grep -E "dp:" | cut -c1-400
STEPS=30 bash scripts/ab_env2.sh 2 "ALM_DEFER_GROUPS=1" "ALM_DEFER_GROUPS=2" "ALM_DEFER_GROUPS=3" > gpurun_out/${tag}_ab_groups.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab_groups.log | cut -c1-200
echo "total t=$((SECONDS-t0))"
