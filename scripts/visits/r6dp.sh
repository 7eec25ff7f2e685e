#!/bin/bash
# round 6: the N > 1 bench path end to end on ONE GPU (two ranks share cuda:0, gloo transport): the self-launch, the engine with the (L-1, 1) cut, the view contract
export PYTHONUNBUFFERED=1 TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
ALM_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 8 --warmup 3 --no-cpu-baseline --no-optimizer-leg > gpurun_out/r6dp_selflaunch.log 2>&1
echo "rc=$?"; grep -E "bench rank|Error|error|Traceback" gpurun_out/r6dp_selflaunch.log | cut -c1-400 | head; tail -n 1 gpurun_out/r6dp_selflaunch.log | cut -c1-1800
ALM_BENCH_SHARE_GPU=1 ALM_DP_GROUP_SIZES=even timeout 900 python bench.py --gpus 2 --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/r6dp_selflaunch_even_opt.log 2>&1
echo "rc=$?"; tail -n 1 gpurun_out/r6dp_selflaunch_even_opt.log | cut -c1-600
