#!/bin/bash
# round 6: 320 x 256 NT tile for ragged token counts (M = 16392: N = 1024 outputs as 208 tiles in ONE round) -- tile tests, step A/B (ALM_GEMM_T320_COST=0 = off)
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "gemm_nt_tile_configs" > gpurun_out/r6t320_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/r6t320_tests.log
run() { env $3 timeout 400 python bench.py --config $1 --steps $4 --warmup 4 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$1 [$2]', d['ms_per_step'], d['roofline']['all_gemm_launches']['by_kind_ms'], 'loss', d['loss'])"; }
for r in 1 2 3; do
  run fine2049 "off" "ALM_GEMM_T320_COST=0" 30
  run fine2049 "t320 1.22" "X=1" 30
  run fine2049 "t320 1.30" "ALM_GEMM_T320_COST=130" 30
done > gpurun_out/r6t320_ab.log 2>&1
for r in 1 2; do
  run e2e_config5 "off" "ALM_GEMM_T320_COST=0" 8
  run e2e_config5 "t320 1.22" "X=1" 8
done >> gpurun_out/r6t320_ab.log 2>&1
cat gpurun_out/r6t320_ab.log
