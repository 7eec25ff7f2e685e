#!/bin/bash
# round 6: owned embedding scatter with a workgroup-shared scan -- kernel tests, determinism test, e2e_config5 / headline A/B against the previous build
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_opwise_model.py tests/test_gpu_defaults.py -m gpu -q --tb=short -x -k "embed or scatter or deterministic" > gpurun_out/r6emb_tests.log 2>&1
echo "tests rc=$?"; tail -n 3 gpurun_out/r6emb_tests.log
B=$PWD/scripts/ubench/bin
for r in 1 2; do
  for lib in $B/libaudiolm_hip_prev.so ""; do
    for cf in e2e_config5 coarse2048; do
    ALM_LIB_PATH=$lib timeout 600 python bench.py --config $cf --steps 10 --warmup 3 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$cf [${lib##*/}]', d['ms_per_step'], 'loss', d['loss'])"
    done
  done
done 2>&1 | tee gpurun_out/r6emb_ab.log
