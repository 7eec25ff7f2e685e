#!/bin/bash
# round 4, visit E: rocprofv3 kernel trace of the bench step (side stream off and on) -> per-kernel stats + idle-gap analysis; optimiser-leg diagnostics
tag=${1:-r4e}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
t0=$SECONDS
prof() {   # $1 = ALM_ASYNC_WGRAD
  rm -rf /tmp/prof_$1
  ALM_ASYNC_WGRAD=$1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o r4 -- python bench.py --steps 5 --warmup 2 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_prof_async$1.log 2>&1
  echo "prof async=$1 rc=$? t=$((SECONDS-t0))"
  db=$(find /tmp/prof_$1 -name "*.db" | head -1)
  if [[ -n $db ]]; then
    python scripts/prof_summary.py "$db" gpurun_out/${tag}_kernel_stats_async$1.csv "ALM_ASYNC_WGRAD=$1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --schedule eager --no-cpu-baseline --no-optimizer-leg (incl. priming + warm-up + 1 instrumented step)"
    python scripts/prof_gaps.py "$db" 0.6 | tee gpurun_out/${tag}_gaps_async$1.log
  fi
}
cd /tmp && cd - > /dev/null
prof 0
prof 1
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.log 2>&1
echo "bench rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config'].get('wgrad_path'), d['config'].get('residual_stream_storage')[:40]); print(json.dumps(d.get('with_optimizer')))"
echo "total t=$((SECONDS-t0))"
