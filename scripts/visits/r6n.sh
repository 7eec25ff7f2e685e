#!/bin/bash
# round 6 visit n: data-parallel tests on the GPU + what the weight-gradient grouping costs at N = 1 (even 2 groups vs uneven 4,2 / 5,1 vs one group)
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_dp.py -x -q > gpurun_out/r6n_tests.log 2>&1
tail -3 gpurun_out/r6n_tests.log
STEPS=40 bash scripts/ab_env.sh 3 "ALM_DEFER_GROUPS=1" "ALM_DEFER_GROUPS=2" "ALM_DEFER_GROUPS=2 ALM_DEFER_GROUP_SIZES=4,2" "ALM_DEFER_GROUPS=2 ALM_DEFER_GROUP_SIZES=5,1" "ALM_DEFER_GROUPS=2 ALM_DEFER_GROUP_SIZES=2,4" > gpurun_out/r6n_groups_ab.log 2>&1
cat gpurun_out/r6n_groups_ab.log
