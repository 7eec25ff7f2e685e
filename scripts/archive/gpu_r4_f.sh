#!/bin/bash
# round 4, visit F: allocator history of the fused-optimiser leg; Fine id bookkeeping kernel; e2e_config5 / fine configs with the new bench rows
tag=${1:-r4f}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
t0=$SECONDS
timeout 300 python scripts/debug/opt_allocs.py > gpurun_out/${tag}_opt_allocs.log 2>&1; echo "opt_allocs rc=$? t=$((SECONDS-t0))"; tail -n 25 gpurun_out/${tag}_opt_allocs.log | cut -c1-400
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py -k "fused" -q --tb=short -p no:cacheprovider > gpurun_out/${tag}_fused.log 2>&1; echo "fused rc=$? t=$((SECONDS-t0))"; tail -n 3 gpurun_out/${tag}_fused.log | cut -c1-300
for cfg in fine2049 e2e_config5; do
  timeout 600 python bench.py --config $cfg --steps 10 --warmup 3 --no-optimizer-leg > gpurun_out/${tag}_bench_$cfg.log 2>&1
  echo "bench $cfg rc=$? t=$((SECONDS-t0))"; tail -n 1 gpurun_out/${tag}_bench_$cfg.log | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['config'].get('parity_sample_N'), d.get('parity'), d.get('cpu_baseline', {}).get('value'))
print([(k['kernel'], k['ms_per_step'], k['frac']) for k in d['roofline']['kernels'][:14]])"
done
echo "total t=$((SECONDS-t0))"
