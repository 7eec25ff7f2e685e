#!/bin/bash
# hot-row split, second build (batched listing loads): tests + kernel A/B
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "embed or scatter" --tb=short -p no:cacheprovider > gpurun_out/r6k_tests.log 2>&1
echo "tests rc=$?"; tail -n 5 gpurun_out/r6k_tests.log | cut -c1-250
log=gpurun_out/r6k_scatter_bench.log; : > $log
prev=$PWD/audiolm-pytorch_amd/libaudiolm_hip_prev.so
for r in 1 2; do
  ALM_LIB_PATH=$prev timeout 300 python scripts/embed_scatter_bench.py 2>&1 | grep tokens | tee -a $log
  timeout 300 python scripts/embed_scatter_bench.py 2>&1 | grep tokens | tee -a $log
done
rm -rf /tmp/prof_k; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_k -o k -- python scripts/embed_scatter_bench.py > /dev/null 2>&1
db=$(find /tmp/prof_k -name "*.db" | head -1); [[ -n $db ]] && python scripts/prof_summary.py "$db" gpurun_out/r6k_scatter_kernels.csv "embed_scatter_bench" | tail -2
head -8 gpurun_out/r6k_scatter_kernels.csv | cut -c1-200
