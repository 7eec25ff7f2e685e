#!/bin/bash
# round 4, visit G: batched hyper-connection finish (tests + in-step A/B), optimiser leg with priming + allocator trace
tag=${1:-r4g}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
run() { name=$1; shift; timeout 1200 python -X faulthandler -m pytest "$@" -q --tb=short --timeout 900 -p no:cacheprovider > gpurun_out/${tag}_${name}.log 2>&1; echo "$name rc=$? t=$((SECONDS-t0))"; tail -n 4 gpurun_out/${tag}_${name}.log | cut -c1-500; }
run hc tests/test_gpu_kernels.py -k "hyper_connection"
run graphed_dp tests/test_gpu_graphed.py tests/test_gpu_dp.py tests/test_gpu_defaults.py
run opwise_b8 tests/test_gpu_opwise.py -k "bf16-8 or coarse-4-bf16-None"
bash scripts/ab_env.sh 3 "ALM_HC_BATCH_FINISH=1" "ALM_HC_BATCH_FINISH=0" 2>&1 | tee gpurun_out/${tag}_ab_finish.log
echo "ab t=$((SECONDS-t0))"
ALM_BENCH_ALLOC_TRACE=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.log 2> gpurun_out/${tag}_bench.err
echo "bench rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['ms_per_step']); print(json.dumps(d.get('with_optimizer')))"
grep -c "alloc-trace" gpurun_out/${tag}_bench.err; grep "alloc-trace" gpurun_out/${tag}_bench.err | head -12 | cut -c1-400
echo "total t=$((SECONDS-t0))"
