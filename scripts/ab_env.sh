#!/bin/bash
# A/B of environment switches inside the headline training step: `scripts/ab_env.sh ROUNDS "ENV=1 ENV2=0" "ENV=0" ...` runs every configuration
# ROUNDS times, interleaved (box-to-box and run-to-run spread is +-2 %), and prints ms/step of each run
export PYTHONUNBUFFERED=1 TMPDIR=/tmp ALM_BENCH_SUPERVISE=0
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    ms=$(env $cfg timeout 300 python bench.py --steps ${STEPS:-40} --warmup 8 --schedule eager --no-cpu-baseline --no-optimizer-leg 2>/dev/null | tail -n 1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['host'].get('eager_issue_ms'))")
    echo "round $r  [$cfg]  $ms"
  done
done
