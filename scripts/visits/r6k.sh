#!/bin/bash
# round 6: kernel traces of the OTHER configurations (looking for shape anomalies: ragged M = 16392, N = 8253 ...), side stream off
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
for cf in fine2049 coarse1024 e2e_config5; do
  rm -rf /tmp/prof_$cf
  ALM_ASYNC_WGRAD=0 timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$cf -o k -- python bench.py --config $cf --steps 3 --warmup 1 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/r6k_prof_$cf.log 2>&1
  db=$(find /tmp/prof_$cf -name "*.db" | head -1)
  [[ -n $db ]] && python scripts/prof_summary.py "$db" gpurun_out/r6k_kernel_stats_$cf.csv "ALM_ASYNC_WGRAD=0 rocprofv3 --kernel-trace -- python bench.py --config $cf --steps 3 --warmup 1 --schedule eager"
done
