#!/bin/bash
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "embed or scatter" --tb=short -p no:cacheprovider > gpurun_out/r6m_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/r6m_tests.log | cut -c1-250
export SCATTER_CASES=headline
prev=$PWD/audiolm-pytorch_amd/libaudiolm_hip_prev.so
for v in prev new; do
  if [[ $v == prev ]]; then export ALM_LIB_PATH=$prev; else unset ALM_LIB_PATH; fi
  rm -rf /tmp/prof_$v; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o k -- python scripts/embed_scatter_bench.py > /dev/null 2>&1
  db=$(find /tmp/prof_$v -name "*.db" | head -1); python scripts/prof_summary.py "$db" gpurun_out/r6m_scatter_kernels_$v.csv "embed_scatter_bench headline uniform ($v)" | tail -1
done
unset ALM_LIB_PATH SCATTER_CASES
log=gpurun_out/r6m_scatter_bench.log; : > $log
for r in 1 2; do
  ALM_LIB_PATH=$prev timeout 300 python scripts/embed_scatter_bench.py 2>&1 | grep tokens | tee -a $log
  timeout 300 python scripts/embed_scatter_bench.py 2>&1 | grep tokens | tee -a $log
done
