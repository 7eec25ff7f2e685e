#!/bin/bash
# round 3, visit B: new 4-wave GEMM tile (14) correctness + A/B, op-wise teacher-forced parity, RCCL 1-rank, host profile
tag=$1
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -n 4 --timeout 300 -k "gemm" > gpurun_out/${tag}_gemm_tests.log 2>&1
echo "gemm tests rc=$? t=$((SECONDS-t0))"; tail -n 12 gpurun_out/${tag}_gemm_tests.log | cut -c1-300
ALM_GEMM_BIG_TILE=14 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -n 4 --timeout 300 -k "gemm" > gpurun_out/${tag}_gemm_tests_tile14.log 2>&1
echo "gemm tests (every big launch on tile 14, NT + TN split-K) rc=$? t=$((SECONDS-t0))"; tail -n 12 gpurun_out/${tag}_gemm_tests_tile14.log | cut -c1-300
timeout 300 python scripts/ab_tiles.py 13 14 11 > gpurun_out/${tag}_ab_tiles.log 2>&1
echo "ab_tiles rc=$? t=$((SECONDS-t0))"; cat gpurun_out/${tag}_ab_tiles.log
export ALM_BENCH_SUPERVISE=0
for bt in 0 14; do
  ALM_GEMM_BIG_TILE=$bt timeout 300 python bench.py --steps 20 --warmup 5 --schedule eager --no-cpu-baseline --no-optimizer-leg > gpurun_out/${tag}_bench_tile$bt.log 2>&1
  echo "bench BIG_TILE=$bt rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_tile$bt.log | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print(d['ms_per_step'], d['host'], r['all_gemm_launches'], [(k['kernel'], k['ms_per_step'], k['frac']) for k in r['kernels']])"
done
unset ALM_BENCH_SUPERVISE
rm -f gpurun_out/r3_opwise_parity.jsonl
timeout 900 python -m pytest tests/test_gpu_opwise.py -m gpu -q --tb=short -s --timeout 800 > gpurun_out/${tag}_opwise.log 2>&1
echo "opwise rc=$? t=$((SECONDS-t0))"; grep -E "op-level|rel-frob|passed|failed|Error|error|> " gpurun_out/${tag}_opwise.log | cut -c1-200 | head -80
timeout 300 python -X faulthandler -m pytest tests/test_gpu_dp.py -m gpu -q --tb=short --timeout 280 -k rccl > gpurun_out/${tag}_rccl.log 2>&1
echo "rccl rc=$? t=$((SECONDS-t0))"; tail -n 15 gpurun_out/${tag}_rccl.log | cut -c1-300
timeout 300 python scripts/host_profile.py > gpurun_out/${tag}_host_profile.log 2>&1
echo "host profile rc=$? t=$((SECONDS-t0))"; head -n 60 gpurun_out/${tag}_host_profile.log | cut -c1-200
echo "total t=$((SECONDS-t0))"
