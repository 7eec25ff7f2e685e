#!/bin/bash
# round 6 visit w: dQ block order = global heaviest-first with a paired first wave (every block count) -- tests + timing against the previous build (r6y) on one box
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bias.py tests/test_gpu_dropout.py -m gpu -q --tb=short -x -k "mqa or attention or bias or dropout" > gpurun_out/r6w_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/r6w_tests.log
B=$PWD/scripts/ubench/bin
for r in 1 2; do
  ALM_LIB_PATH=$B/libaudiolm_hip_r6y.so timeout 600 python scripts/attn_bench.py 1024 2048 2049 2113 8253 16385
  timeout 600 python scripts/attn_bench.py 1024 2048 2049 2113 8253 16385
done > gpurun_out/r6w_dq_order_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r6w_dq_order_ab.log | sed "s#$B/##"
for cf in fine_t2048_q8 e2e_config5 fine2049 coarse2048; do
  for lib in $B/libaudiolm_hip_r6y.so ""; do
    ALM_LIB_PATH=$lib timeout 600 python bench.py --config $cf --steps 10 --warmup 3 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$cf [${lib##*/}]', d['ms_per_step'], {k['kernel']: k['ms_per_step'] for k in d['roofline']['kernels'] if 'mqa' in k['kernel']})"
  done
done 2>&1 | tee -a gpurun_out/r6w_dq_order_ab.log
