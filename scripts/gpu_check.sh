#!/bin/bash
# One GPU-box visit: kernel parity, model parity, smoke, bench (+ rocprofv3 kernel stats).  Logs -> gpurun_out/.
# usage: scripts/gpu_check.sh [tests|kbench|bench|prof|dp2|all]
what=${1:-all}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
python - > gpurun_out/env.log 2>&1 <<'PY'
import torch, os
print('torch', torch.__version__, 'cuda', torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else None)
p = torch.cuda.get_device_properties(0)
print('CUs', p.multi_processor_count, 'mem GB', p.total_memory / 2**30, 'host cores', os.cpu_count())
PY
cat gpurun_out/env.log
if [[ $what == tests || $what == all ]]; then
  timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bias.py tests/test_gpu_codec.py -m gpu -q --tb=short -n 4 --timeout 300 > gpurun_out/kernels.log 2>&1
  echo "kernels rc=$?"; tail -n 60 gpurun_out/kernels.log
  timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -s --timeout 600 > gpurun_out/parity.log 2>&1
  echo "parity rc=$?"; tail -n 80 gpurun_out/parity.log
  timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1
  echo "smoke rc=$?"; tail -n 5 gpurun_out/smoke.log
fi
if [[ $what == kbench || $what == all ]]; then
  timeout 600 python scripts/kbench.py > gpurun_out/kbench.log 2>&1
  echo "kbench rc=$?"; cat gpurun_out/kbench.log
fi
if [[ $what == bench || $what == all ]]; then
  timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench.log 2>&1
  echo "bench rc=$?"; tail -n 12 gpurun_out/bench.log
fi
if [[ $what == dp2 ]]; then
  # two ranks sharing the one GPU over gloo: exercises bench.py's multi-rank control flow (per-layer bucket callbacks, barriers, the
  # instrumented step, the optimizer leg) on a 1-GPU box -- the numbers mean nothing
  ALM_BENCH_SHARE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/dp2.log 2>&1
  echo "dp2 rc=$?"; tail -n 3 gpurun_out/dp2.log | cut -c1-400
fi
if [[ $what == prof || $what == all ]]; then
  export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r1 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer-leg > gpurun_out/prof.log 2>&1
  echo "prof rc=$?"; tail -n 5 gpurun_out/prof.log
  db=$(find gpurun_out/prof -name "*.db" | head -1)
  if [[ -n $db ]]; then python scripts/prof_summary.py "$db" gpurun_out/kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-optimizer-leg (8 steps incl. warm-up + 1 instrumented)"; rm -f "$db"; head -n 30 gpurun_out/kernel_stats.csv | cut -c1-180; fi
fi
