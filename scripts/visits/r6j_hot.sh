#!/bin/bash
# hot-row split of the owned embedding scatter: tests, kernel A/B (previous library vs this one), step A/B of e2e_config5 and the headline
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "embed or scatter" --tb=short -p no:cacheprovider > gpurun_out/r6j_tests.log 2>&1
echo "tests rc=$?"; tail -n 15 gpurun_out/r6j_tests.log | cut -c1-250
log=gpurun_out/r6j_scatter_bench.log; : > $log
prev=$PWD/audiolm-pytorch_amd/libaudiolm_hip_prev.so
for r in 1 2; do
  ALM_LIB_PATH=$prev timeout 300 python scripts/embed_scatter_bench.py 2>&1 | grep tokens | tee -a $log
  timeout 300 python scripts/embed_scatter_bench.py 2>&1 | grep tokens | tee -a $log
  ALM_EMBED_SCATTER_HOT=0 timeout 300 python scripts/embed_scatter_bench.py 2>&1 | grep tokens | sed 's/^default/hot-off/' | tee -a $log
done
log=gpurun_out/r6j_step_ab.log; : > $log
for r in 1 2 3; do
  for cf in e2e_config5 coarse2048; do
    for lib in prev new; do
      if [[ $lib == prev ]]; then export ALM_LIB_PATH=$prev; else unset ALM_LIB_PATH; fi
      ms=$(timeout 600 python bench.py --config $cf --steps 10 --warmup 3 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('loss'))")
      echo "round $r $cf [$lib] $ms" | tee -a $log
    done
  done
done
