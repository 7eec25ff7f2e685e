#!/bin/bash
# round 6, visit a: in-launch split-K NT (tests, cost-model measurements, step A/B with the workspace off / on on the headline and configs[1])
tag=${1:-r6a}
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -x -k "inlaunch or workspace or group2 or gemm_nt" -s > gpurun_out/${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -n 25 gpurun_out/${tag}_tests.log
timeout 600 python scripts/ab_nt_inl.py 8192 16384 2048 > gpurun_out/${tag}_nt_inl_cost.log 2>&1
echo "cost rc=$?"; cat gpurun_out/${tag}_nt_inl_cost.log
for r in 1 2; do
  for cfg in coarse1024 coarse2048; do
    for e in "ALM_GEMM_NT_WS=0 ALM_GEMM_GROUP2_BIG=0" "ALM_GEMM_NT_WS=1"; do
      ms=$(env $e timeout 300 python bench.py --config $cfg --steps 30 --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('roofline') or {}).get('all_gemm_launches', {}).get('by_kind_ms'), d['host']['eager_issue_ms'])")
      echo "round $r $cfg [$e] $ms"
    done
  done
done > gpurun_out/${tag}_step_ab.log 2>&1
cat gpurun_out/${tag}_step_ab.log
