#!/bin/bash
tag=${1:-r5e}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 900 python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_optimizer.py tests/test_gpu_graphed.py -m gpu -q --tb=short --timeout 600 -p no:cacheprovider \
  -k "pack or graphed_step or fused_adam" > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/${tag}_tests.log | cut -c1-300
STEPS=30 bash scripts/ab_env2.sh 3 "ALM_PACK_WIDE=1 ALM_PACK_ALL=1" "ALM_PACK_WIDE=0 ALM_PACK_ALL=0" "ALM_PACK_WIDE=1 ALM_PACK_ALL=0" > gpurun_out/${tag}_ab.log 2>&1
echo "ab t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab.log | cut -c1-100
export ALM_BENCH_SUPERVISE=0
rm -rf /tmp/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python bench.py --steps 5 --warmup 2 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_prof.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_kernel_stats.csv
grep -i "pack" gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
echo "total t=$((SECONDS-t0))"
