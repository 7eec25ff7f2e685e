#!/bin/bash
# like ab_env.sh, but also prints the instrumented step's per-kind GEMM times (roofline.all_gemm_launches.by_kind_ms) and the dominant kernel's average
# launch: `scripts/ab_env2.sh ROUNDS "ENV=1" "ENV=0" ...`
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    out=$(env $cfg timeout 300 python bench.py --steps ${STEPS:-40} --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d.get('roofline') or {}
kk = {k['name'] if 'name' in k else k.get('kernel', '?')[:24]: k for k in r.get('kernels', [])} if isinstance(r.get('kernels'), list) else {}
print(d['ms_per_step'], 'nt_big_us', r.get('avg_launch_us'), 'frac', r.get('frac'), 'gemm_ms', (r.get('all_gemm_launches') or {}).get('by_kind_ms'), 'instr_ms', r.get('instrumented_ms_per_step'))")
    echo "round $r  [$cfg]  $out"
  done
done
