#!/bin/bash
# round 6 visit v: odd block counts in the dQ kernel's XCD block map (N = 2049: configs[2]) -- tests + timing against the previous build on one box
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "mqa_attention" > gpurun_out/r6v_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/r6v_tests.log
B=$PWD/scripts/ubench/bin
for r in 1 2; do
  ALM_LIB_PATH=$B/libaudiolm_hip_r6y.so timeout 600 python scripts/attn_bench.py 1024 2048 2049 2113 8253
  timeout 600 python scripts/attn_bench.py 1024 2048 2049 2113 8253
done > gpurun_out/r6v_odd_blocks_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r6v_odd_blocks_ab.log | sed "s#$B/##"
for r in 1 2; do
  for lib in $B/libaudiolm_hip_r6y.so ""; do
    ALM_LIB_PATH=$lib timeout 300 python bench.py --config fine2049 --steps 30 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('fine2049 [${lib##*/}]', d['ms_per_step'], {k['kernel']: k['ms_per_step'] for k in d['roofline']['kernels'] if 'mqa' in k['kernel']})"
  done
done 2>&1 | tee -a gpurun_out/r6v_odd_blocks_ab.log
