#!/bin/bash
# round 3, visit D: A/B of a kernel change inside the training step: tests named in $TESTS, then bench + rocprofv3 kernel stats
tag=$1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_kernels.py} -m gpu -q --tb=short --timeout 600 ${TEST_ARGS} > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$? t=$((SECONDS-t0))"; tail -n 12 gpurun_out/${tag}_tests.log | cut -c1-300
export ALM_BENCH_SUPERVISE=0
timeout 300 python bench.py --steps 20 --warmup 5 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_bench.log 2>&1
echo "bench rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['ms_per_step'], d['host'], r['all_gemm_launches'])
for k in r['kernels']: print('  ', k['kernel'], k['launches_per_step'], k['avg_launch_us'], k['ms_per_step'], k['achieved'], k['unit'], k['frac'])"
rm -rf /tmp/prof_d
ALM_ASYNC_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_d -o r3 -- python bench.py --steps 5 --warmup 2 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_prof.log 2>&1
echo "prof rc=$? t=$((SECONDS-t0))"
db=$(find /tmp/prof_d -name "*.db" | head -1)
if [[ -n $db ]]; then python scripts/prof_summary.py "$db" gpurun_out/${tag}_kernel_stats_bf16_serial.csv "ALM_ASYNC_WGRAD=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --schedule eager --no-cpu-baseline --no-optimizer-leg (3 + 10 priming + 2 warm-up + 5 timed + 2 instrumented/host steps)"; head -n 24 gpurun_out/${tag}_kernel_stats_bf16_serial.csv | cut -c1-150; fi
echo "total t=$((SECONDS-t0))"
