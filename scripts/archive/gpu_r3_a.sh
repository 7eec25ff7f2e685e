#!/bin/bash
# round 3, visit A: the new parity tests (rounding-matched oracle, real-reference digests), RCCL 1-rank, graph fix, bench line with roofline.kernels, kernel stats
tag=$1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
rm -f gpurun_out/r3_fullsize_parity.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -s --timeout 600 > gpurun_out/${tag}_parity.log 2>&1
echo "parity rc=$? t=$((SECONDS-t0))"; grep -E "rounding-matched|logits rel-frob|  grad |passed|failed|Error" gpurun_out/${tag}_parity.log | cut -c1-200 | tail -n 60
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -s --timeout 900 > gpurun_out/${tag}_fullsize.log 2>&1
echo "fullsize rc=$? t=$((SECONDS-t0))"; grep -E "loss ours|ROUNDING|grads: worst|REAL reference|logits sample|worst grad|passed|failed|Error" gpurun_out/${tag}_fullsize.log | cut -c1-260 | tail -n 80
timeout 600 python -X faulthandler -m pytest tests/test_gpu_dp.py tests/test_gpu_graphed.py -m gpu -q --tb=short --timeout 500 > gpurun_out/${tag}_dp_graphed.log 2>&1
echo "dp+graphed rc=$? t=$((SECONDS-t0))"; tail -n 25 gpurun_out/${tag}_dp_graphed.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/${tag}_bench_full.log 2>&1
echo "bench full rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_full.log | cut -c1-6000
export ALM_BENCH_SUPERVISE=0
rm -rf /tmp/prof_a
ALM_ASYNC_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_a -o r3 -- python bench.py --steps 5 --warmup 2 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_prof.log 2>&1
echo "prof rc=$? t=$((SECONDS-t0))"
db=$(find /tmp/prof_a -name "*.db" | head -1)
if [[ -n $db ]]; then python scripts/prof_summary.py "$db" gpurun_out/${tag}_kernel_stats_bf16_serial.csv "ALM_ASYNC_WGRAD=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --schedule eager --no-cpu-baseline --no-optimizer-leg (3 + 10 priming + 2 warm-up + 5 timed + 2 instrumented/host steps)"; head -n 40 gpurun_out/${tag}_kernel_stats_bf16_serial.csv | cut -c1-160; fi
for m in flat nested; do
  timeout 120 python scripts/ubench/nested_fork_capture.py $m > gpurun_out/${tag}_nested_fork_$m.log 2>&1
  echo "nested_fork $m rc=$? t=$((SECONDS-t0))"; tail -n 6 gpurun_out/${tag}_nested_fork_$m.log | cut -c1-200
done
echo "total t=$((SECONDS-t0))"
